#!/usr/bin/env python
"""Round 6: where the unit boundary of K17's persistent form spends its time (s_memtime stamps, -DFAR_WINO_TIMING3).
  python tools/wino_persist_timing.py --build   (CPU) far_amd/lib/exp/libfar_winot3.so
  FAR_HIP_LIB=far_amd/lib/exp/libfar_winot3.so python tools/wino_persist_timing.py [--mode 0|2]   (GPU)"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXP_LIB = os.path.join(ROOT, 'far_amd', 'lib', 'exp', 'libfar_winot3.so')
NAMES = ['loop top', 'prologue landed + barrier', 'transform(0)', 'K loop', 'drain + barrier', 'overflow check, Z write, residual requests',
         'Z barrier', 'Z reads + residual wait', 'LDS-free barrier', 'setup + prologue requests', 'output + stores issued']


def build():
    from far_amd import build as B
    B.build(verbose=False)
    os.makedirs(os.path.dirname(EXP_LIB), exist_ok=True)
    objs = []
    for src in B.sources():
        base = os.path.basename(src)
        obj = os.path.join(B.LIBDIR, base[:-4] + '.o')
        if base == 'conv_wino_f16s.hip':
            obj = os.path.join(os.path.dirname(EXP_LIB), 'conv_wino_f16s.t3.o')
            subprocess.check_call([B.HIPCC] + B.FLAGS + ['-DFAR_WINO_TIMING3', '-c', src, '-o', obj])
        objs.append(obj)
    subprocess.check_call([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', EXP_LIB] + objs)
    print(EXP_LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--mode', type=int, default=0)
    a = ap.parse_args()
    if a.build:
        return build()
    import numpy as np
    import torch
    from far_amd import _lib, ops
    lib = _lib.load()
    lib.far_set_tuning(14, a.mode)
    g = torch.Generator(device='cuda').manual_seed(1)
    for (H, W, ci, co) in ((240, 320, 128, 128), (240, 320, 208, 208)):
        x = torch.randn(64, H, W, ci, device='cuda', generator=g).relu_()
        w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5
        pw = ops.PackedWino(w, torch.ones(co, device='cuda'), torch.zeros(co, device='cuda'))
        for _ in range(3):
            ops.conv3x3_wino(x, pw, act='relu')
        torch.cuda.synchronize()
        buf = np.zeros((256, 2, 4, 16), dtype=np.uint64)
        rc = ctypes.CDLL(os.environ['FAR_HIP_LIB']).far_wino_timing3_dump(buf.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        st = buf.astype(np.int64)
        print(f'## {ci}->{co} @{H}x{W} x64, tuning 14={a.mode}: s_memtime ticks (100 MHz) x 23 ~ cycles at 2.3 GHz; mean over 256 workgroups x units 1..4')
        for wv in (0, 1):
            d = np.diff(st[:, wv, :, :11], axis=-1).astype(np.float64)          # (256, 4, 10)
            unit = (st[:, wv, 1:, 0] - st[:, wv, :-1, 0]).astype(np.float64)     # loop top to loop top
            print(f'  wave {4 * wv}: unit period {unit.mean() * 23:.0f} cycles (min {unit.min() * 23:.0f}, max {unit.max() * 23:.0f})')
            for i in range(10):
                print(f'     {NAMES[i + 1]:48s} {d[:, :, i].mean() * 23:8.0f}')
            tail = (st[:, wv, 1:, 0] - st[:, wv, :-1, 10]).astype(np.float64)
            print(f'     {"(stores issued -> next loop top)":48s} {tail.mean() * 23:8.0f}')
    lib.far_set_tuning(14, 0)


if __name__ == '__main__':
    main()
