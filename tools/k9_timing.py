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
if os.environ.get('K9_STAGGER'):
    lib.far_set_tuning(2, int(os.environ['K9_STAGGER']))
tk = (ctypes.c_ulonglong * 2)()
lib.far_k9_tick_probe.argtypes = [ctypes.c_ulonglong, ctypes.c_void_p]
lib.far_k9_tick_probe(200_000_000, tk)
print(f's_memtime: {tk[0]} ticks in {tk[1]} ticks of the 100 MHz clock -> {tk[0] / tk[1] * 100:.0f} MHz (idle GPU, one wave)')
which = [a for a in sys.argv[1:] if not a.startswith('-')] or ['linear']
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)


def report(name, nblocks, wg_per_cu):
    nb = min(nblocks, 65536)
    buf = np.zeros((nb, 12), dtype=np.uint64)
    assert lib.far_k9_timing_dump(buf.ctypes.data, nb) == 0
    t = buf[:, :8].astype(np.int64)
    t0 = t[:, 0].min()
    rt = (buf[:, 9] - buf[:, 8]).astype(np.int64)
    print(f'  lifetime on the 100 MHz clock: mean {rt.mean() * 10:.0f} ns -> shader clock {(t[:, 4] - t[:, 0]).mean() / (rt.mean() * 10) * 1e3:.0f} MHz during the kernel')
    pro, loop, epi, life = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 0]
    epi0, drain = t[:, 5] - t[:, 2], t[:, 4] - t[:, 3]
    epiw = t[:, 6] - t[:, 2]
    spans = [int(t[x::8, 4].max() - t[x::8, 0].min()) for x in range(8)]     # every XCD has its own counter
    span = max(spans)
    print(f'  per-XCD span (ticks): {spans}')
    print(f'## {name}: {nblocks} workgroups ({nb} sampled), kernel span {span} ticks')
    for nm, v in (('prologue', pro), ('K loop', loop), ('epilogue issue', epi), (' (first tile)', epi0), (' (its LDS write)', epiw), (' (LN: to stats done)', t[:, 7] - t[:, 2]), ('store drain', drain), ('lifetime', life)):
        print(f'  {nm:15s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}   {100 * v.mean() / life.mean():5.1f} %')
    # resident workgroups over time
    ev = np.concatenate([np.stack([t[:, 0], np.ones(nb, np.int64)], 1), np.stack([t[:, 4], -np.ones(nb, np.int64)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind='stable')]
    occ = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0])
    print(f'  mean resident workgroups {float((occ[:-1] * dt).sum()) / max(1, int(dt.sum())):.1f} of {256 * wg_per_cu} slots'
          f' (sampled blocks only)')


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
    elif w == 'linear_ln':
        r = torch.randn(1, 1, 64 * 4800, 256, device=dev, generator=g)
        pl = ops.PackedConv(torch.randn(256, 256, device=dev, generator=g) * 0.05)
        gm, bt = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        for _ in range(3):
            ops.conv_nhwc(r, pl, ln=(gm, bt, 1e-5), post_residual=r)
        torch.cuda.synchronize()
        report('linear 256->256 + LayerNorm + residual', 64 * 4800 // 128, 2)
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
