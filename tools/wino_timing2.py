"""Where the waves of K17 wait inside k-steps 2 and 3 (library built with -DFAR_WINO_TIMING2, passed as FAR_HIP_LIB): LDS-resident
stamps, so the request queue is undisturbed.  Usage: FAR_HIP_LIB=... python tools/wino_timing2.py"""
import ctypes
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from far_amd import _lib, ops

lib = _lib.load()
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(64, 240, 320, 128, device='cuda', generator=g).relu_()
w = torch.randn(128, 128, 3, 3, device='cuda', generator=g) * (2.0 / (128 * 9)) ** 0.5
pw = ops.PackedWino(w, torch.ones(128, device='cuda'), torch.zeros(128, device='cuda'))
for _ in range(3):
    ops.conv3x3_wino(x, pw, act='relu')
torch.cuda.synchronize()
NB = 4096
buf = np.zeros((NB, 8, 16), dtype=np.uint64)
fn = lib.far_wino_timing2_dump
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data_as(ctypes.c_void_p), NB) == 0
t = buf[1024:].astype(np.int64)
names = ['work (even)', 'request wait', 'barrier', 'work (odd)', 'request wait', 'barrier']
for kk in (0, 1):
    print(f'k-step {2 + kk}:')
    for w_ in range(8):
        s = t[:, w_, 8 * kk:8 * kk + 7]
        d = np.diff(s, axis=1).mean(0)
        print(f'  wave {w_}: ' + '  '.join(f'{n} {v:.0f}' for n, v in zip(names, d)) + f'   k-step {np.mean(s[:, 6] - s[:, 0]):.0f}')
