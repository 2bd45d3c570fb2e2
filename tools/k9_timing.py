"""Development aid: where a K9 workgroup's lifetime goes (s_memtime stamps compiled in with -DFAR_K9_TIMING).

Build (here, no GPU needed):   python tools/k9_timing.py --build
Run (GPU box):                 python tools/k9_timing.py [linear|conv196|conv256]
The instrumented library is far_amd/lib/libfar_hip_timing.so; the product library is untouched.
"""
import ctypes, os, subprocess, sys
sys.path.insert(0, '.')
from far_amd import build as B, _lib

TLIB = os.path.join(B.LIBDIR, 'libfar_hip_timing.so')
if '--build' in sys.argv:
    B.build(verbose=False)
    obj = os.path.join(B.LIBDIR, 'conv_igemm_f16s_timing.o')
    subprocess.check_call([B.HIPCC] + B.FLAGS + ['-DFAR_K9_TIMING', '-c', os.path.join(B.CSRC, 'conv_igemm_f16s.hip'), '-o', obj])
    objs = [os.path.join(B.LIBDIR, os.path.basename(s)[:-4] + '.o') for s in B.sources()]
    objs = [obj if o.endswith('conv_igemm_f16s.o') else o for o in objs]
    subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', TLIB] + objs)
    print(TLIB)
    sys.exit(0)

import numpy as np
import torch
_lib.LIB_PATH = TLIB
from far_amd import ops
lib = _lib.load()
lib.far_k9_timing_dump.restype = ctypes.c_int
lib.far_k9_timing_dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
tk = (ctypes.c_ulonglong * 2)()
lib.far_k9_tick_probe.argtypes = [ctypes.c_ulonglong, ctypes.c_void_p]
lib.far_k9_tick_probe(200_000_000, tk)
print(f's_memtime: {tk[0]} ticks in {tk[1]} ticks of the 100 MHz clock -> {tk[0] / tk[1] * 100:.0f} MHz (idle GPU, one wave)')
which = [a for a in sys.argv[1:] if not a.startswith('-')] or ['linear']
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)


def report(name, nblocks, wg_per_cu, ln=False):
    """Stamps (s_memtime = shader cycles) of wave 0 of every workgroup: 0 entry, 1 prologue done, 2 K loop done,
    3 epilogue issued, 4 stores acknowledged, 5 first epilogue tile done, 6 its LDS transpose written,
    7 (LayerNorm) its statistics done; 8 / 9 = s_memrealtime (100 MHz) at entry / exit."""
    nb = min(nblocks, 65536)
    buf = np.zeros((nb, 12), dtype=np.uint64)
    assert lib.far_k9_timing_dump(buf.ctypes.data, nb) == 0
    t = buf[:, :8].astype(np.int64)
    life = t[:, 4] - t[:, 0]
    rt = (buf[:, 9] - buf[:, 8]).astype(np.int64)
    print(f'## {name}: {nblocks} workgroups, {wg_per_cu} per CU')
    print(f'  shader clock during the kernel: {life.mean() / (rt.mean() * 10) * 1e3:.0f} MHz '
          f'(workgroup lifetime {life.mean():.0f} cycles = {rt.mean() * 10:.0f} ns on the 100 MHz clock)')
    rows = [('prologue', t[:, 1] - t[:, 0]), ('K loop', t[:, 2] - t[:, 1]), ('epilogue', t[:, 3] - t[:, 2]),
            ('  first tile', t[:, 5] - t[:, 2]), ('    its LDS transpose', t[:, 6] - t[:, 2])]
    if ln:
        rows.append(('    its LayerNorm statistics', t[:, 7] - t[:, 6]))
    rows += [('store acknowledgement', t[:, 4] - t[:, 3]), ('lifetime', life)]
    for nm, v in rows:
        print(f'  {nm:30s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles  {100 * v.mean() / life.mean():5.1f} %')


for w in which:
    if w == 'linear':
        r = torch.randn(1, 1, 64 * 4800, 256, device=dev, generator=g)
        pl = ops.PackedConv(torch.randn(256, 256, device=dev, generator=g) * 0.05)
        for _ in range(3):
            ops.conv_nhwc(r, pl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv_nhwc(r, pl); e1.record(); torch.cuda.synchronize()
        print(f'event-timed launch {e0.elapsed_time(e1) * 1e3:.1f} us')
        report('linear 256->256, 307200 rows', 64 * 4800 // 128, 2)
    elif w == 'linear_small':
        r = torch.randn(1, 1, 4800, 256, device=dev, generator=g)
        pl = ops.PackedConv(torch.randn(256, 256, device=dev, generator=g) * 0.05)
        for _ in range(3):
            ops.conv_nhwc(r, pl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv_nhwc(r, pl); e1.record(); torch.cuda.synchronize()
        print(f'event-timed launch {e0.elapsed_time(e1) * 1e3:.1f} us')
        report('linear 256->256, 4800 rows (half-height tiles)', 4800 // 64, 1)
    elif w == 'linear_ln':
        r = torch.randn(1, 1, 64 * 4800, 256, device=dev, generator=g)
        pl = ops.PackedConv(torch.randn(256, 256, device=dev, generator=g) * 0.05)
        gm, bt = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        for _ in range(3):
            ops.conv_nhwc(r, pl, ln=(gm, bt, 1e-5), post_residual=r)
        torch.cuda.synchronize()
        report('linear 256->256 + LayerNorm + residual', 64 * 4800 // 128, 2, ln=True)
    elif w in ('conv196', 'conv256', 'conv128', 'conv128res'):
        C, H, W = {'conv196': (196, 240, 320), 'conv256': (256, 120, 160), 'conv128': (128, 240, 320), 'conv128res': (128, 240, 320)}[w]
        x = torch.randn(64, H, W, C, device=dev, generator=g).relu_()
        pc = ops.PackedConv(torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.03, torch.ones(C, device=dev), torch.zeros(C, device=dev))
        for _ in range(2):
            ops.conv_nhwc(x, pc, act='relu', residual=x if w.endswith('res') else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv_nhwc(x, pc, act='relu', residual=x if w.endswith('res') else None); e1.record(); torch.cuda.synchronize()
        print(f'event-timed launch {e0.elapsed_time(e1) * 1e3:.1f} us')
        px = 256 if C <= 128 else 128
        report(f'3x3 {C}->{C} @{H}x{W} x64', 64 * ((H + (px // 16) - 1) // (px // 16)) * (W // 16), 2)
        del x
