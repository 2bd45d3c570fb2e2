"""Diagnostic: K1 / K2 forward timings at the bench shapes (32 pairs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import ops          # noqa: E402


def t_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator(device='cuda').manual_seed(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
q = torch.randn(n * 8, 4800, 64, device='cuda', generator=g)
k = torch.randn(n * 8, 4800, 64, device='cuda', generator=g)
v = torch.randn(n * 8, 4800, 64, device='cuda', generator=g)
pos = torch.rand(4800, 6, device='cuda', generator=g)
print(f'K2 far_emm_pv_f16s all passes, {n} pairs: {t_ms(lambda: ops._emm_pv(q, k, v, pos, 0.125)):.3f} ms')
f0 = 1.2 * torch.randn(n, 4800, 256, device='cuda', generator=g)
f1 = f0[:, torch.randperm(4800, device='cuda', generator=g)] + 0.1 * torch.randn(n, 4800, 256, device='cuda', generator=g)
print(f'K1 far_coarse_match_f16s all passes, {n} pairs: {t_ms(lambda: ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, variant="f16s")):.3f} ms')
