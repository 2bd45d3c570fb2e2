"""Per-workgroup s_memtime timeline of K17 (library built with -DFAR_WINO_TIMING, passed as FAR_HIP_LIB).
Usage: FAR_HIP_LIB=/path/libfar_timing.so python tools/wino_timing.py [H W Cin Cout]"""
import ctypes
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from far_amd import _lib, ops

lib = _lib.load()
H, W, ci, co = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (240, 320, 128, 128)
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(64, H, W, ci, device='cuda', generator=g).relu_()
w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5
pw = ops.PackedWino(w, torch.ones(co, device='cuda'), torch.zeros(co, device='cuda'))
for _ in range(3):
    ops.conv3x3_wino(x, pw, act='relu')
torch.cuda.synchronize()
NB = 4096
buf = np.zeros((NB, 8, 64), dtype=np.uint64)
fn = lib.far_wino_timing_dump
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data_as(ctypes.c_void_p), NB) == 0
t = buf[1024:].astype(np.int64)
nk = (ci + 15) // 16
nki = min(nk, 14)
for wsel, name in [(w, f'wave {w} (xi {w >> 1}, tile block {w & 1}: multiplies in {"even" if w < 4 else "odd"} intervals)') for w in range(8)]:
    s = t[:, wsel, :]
    tot = s[:, 63] - s[:, 0]
    print(f'{name}: lifetime {tot.mean():.0f} cycles (min {tot.min()}, max {tot.max()})')
    print(f'  prologue {np.mean(s[:, 1] - s[:, 0]):.0f}')
    work_e, wait_e, work_o, wait_o = [], [], [], []
    for k in range(nki):
        prev = s[:, 1] if k == 0 else s[:, 5 + 4 * (k - 1)]
        work_e.append(np.mean(s[:, 2 + 4 * k] - prev)); wait_e.append(np.mean(s[:, 3 + 4 * k] - s[:, 2 + 4 * k]))
        work_o.append(np.mean(s[:, 4 + 4 * k] - s[:, 3 + 4 * k])); wait_o.append(np.mean(s[:, 5 + 4 * k] - s[:, 4 + 4 * k]))
    print('  even intervals: work ' + ' '.join(f'{v:.0f}' for v in work_e))
    print('                  wait ' + ' '.join(f'{v:.0f}' for v in wait_e))
    print('  odd intervals:  work ' + ' '.join(f'{v:.0f}' for v in work_o))
    print('                  wait ' + ' '.join(f'{v:.0f}' for v in wait_o))
    print(f'  K loop total {np.mean(s[:, 60] - s[:, 1]):.0f}  Z exchange {np.mean(s[:, 61] - s[:, 60]):.0f}  output {np.mean(s[:, 62] - s[:, 61]):.0f}  store drain {np.mean(s[:, 63] - s[:, 62]):.0f}')
