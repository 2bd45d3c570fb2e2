#!/usr/bin/env python
"""Event-timed K14 (attention block) and K13 (MLP block) of the fine-level LoFTR layers at the bench shapes (60 148 matches per 32
pairs: 'self' = both images stacked = 120 296 windows, 'cross' = 60 148), current pipelines against the round-3 ones
(round 5: far_set_tuning 11; the variants were removed from the library after this tool showed the two-workgroups-per-CU forms of K14 to be run-to-run non-deterministic at this size -- profiles/r05_fine_level.txt; the tool now times the shipped kernels and repeats them for bit-equality), with a bit-equality check between the two.   python tools/fine_time.py [--windows 60148]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ev(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--windows', type=int, default=60148)
    a = ap.parse_args()
    from far_amd import _lib, ops
    lib = _lib.load()
    D, H = 128, 8
    g = torch.Generator(device='cuda').manual_seed(5)
    ws = [torch.randn(D, D, device='cuda', generator=g) / 11 for _ in range(4)]
    gam, bet = torch.rand(D, device='cuda', generator=g) + 0.5, torch.randn(D, device='cuda', generator=g) * 0.1
    w0 = torch.randn(2 * D, 2 * D, device='cuda', generator=g) / 16
    w2 = torch.randn(D, 2 * D, device='cuda', generator=g) / 16
    pa, pm = ops.PackedAttn(*ws), ops.PackedMlp(w0, w2)
    for n in (2 * a.windows, a.windows):
        x = torch.randn(n, 25, D, device='cuda', generator=g)
        s = torch.randn(n, 25, D, device='cuda', generator=g)
        for name in ('K14', 'K13'):
            fn = (lambda: ops.attn_block(x, s, pa, H, gam, bet, 1e-5)) if name == 'K14' else (lambda: ops.mlp_fused(x, s, pm, gam, bet, 1e-5))
            y = fn()
            t = ev(fn)
            again = [fn() for _ in range(4)]
            bad = [int(((a_ - y).abs().flatten(1).max(1).values > 0).sum()) for a_ in again]
            print(f'windows {n}: {name} {t:.3f} ms; windows differing from the first launch in 4 more launches: {bad}', flush=True)


if __name__ == '__main__':
    main()
