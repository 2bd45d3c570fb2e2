"""K1 (split-fp16) per pass on the bench workload's coarse features (32 pairs), with and without the match-pass prescreen.
Usage: python tools/k1_time.py"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from far_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device='cuda').manual_seed(1)
L, C, n = 4800, 256, 32
f0 = 1.2 * torch.randn(n, L, C, device='cuda', generator=g)
f1 = f0[:, torch.randperm(L, device='cuda', generator=g)] + 0.1 * torch.randn(n, L, C, device='cuda', generator=g)
for off in (0, 1, 0, 1):
    lib.far_set_tuning(10, off)
    t = min(bench.event_time_ms(lambda: ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, variant='f16s'), iters=5, warm=2) for _ in range(3))
    print(f'prescreen {"off" if off else "on "}: far_coarse_match_f16s all passes {t:.3f} ms', flush=True)
lib.far_set_tuning(10, 0)
