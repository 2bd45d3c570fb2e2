"""Where the waves of K18 spend the units of k-steps 2 and 3 (library built with -DFAR_W1D_TIMING, passed as FAR_HIP_LIB).
Usage: FAR_HIP_LIB=... python tools/w1d_timing.py"""
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
pw = ops.PackedWino1d(w, torch.ones(128, device='cuda'), torch.zeros(128, device='cuda'))
for _ in range(3):
    ops.conv3x3_wino1d(x, pw, act='relu')
torch.cuda.synchronize()
NB = 4096
buf = np.zeros((NB, 8, 32), dtype=np.uint64)
fn = lib.far_w1d_timing_dump
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data_as(ctypes.c_void_p), NB) == 0
t = buf[1024:].astype(np.int64)
for w_ in range(8):
    s = t[:, w_, :24].reshape(-1, 6, 4)
    role = 'P: multiply, then transform' if w_ < 4 else 'Q: transform, then multiply'
    first = (s[:, 1:, 0] - s[:, :-1, 3]).mean(0)
    second = (s[:, :, 2] - s[:, :, 0]).mean(0)
    wait = (s[:, :, 3] - s[:, :, 2]).mean(0)
    print(f'wave {w_} ({role}); units of k-steps 2, 3 (ky 0 1 2 0 1 2); s_memtime ticks (2.2 per ns); unit {np.mean(s[:, 5, 3] - s[:, 0, 3]) / 5:.0f}')
    print('   first half   ' + '    - ' + ' '.join(f'{v:5.0f}' for v in first))
    print('   second half  ' + ' '.join(f'{v:5.0f}' for v in second))
    print('   wait+barrier ' + ' '.join(f'{v:5.0f}' for v in wait))
