"""Diagnostic: cost of the fused LayerNorm epilogue of K9's Linear mode relative to the plain epilogue (same run, same box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import ops          # noqa: E402


def t_ms(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


torch.manual_seed(0)
for rows, cin, cout in ((153600, 256, 256), (153600, 512, 256), (1525000, 128, 128), (1525000, 256, 128)):
    w = torch.randn(cout, cin, device='cuda') / cin ** 0.5
    pk = ops.PackedConv(w, split=True)
    x = torch.randn(1, rows, cin, device='cuda')
    r = torch.randn(1, rows, cout, device='cuda')
    out = torch.empty(1, rows, cout, device='cuda')
    g, b = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda')
    t0 = t_ms(lambda: ops.linear_f16s(x, pk, out=out))
    t1 = t_ms(lambda: ops.linear_f16s(x, pk, ln=(g, b, 1e-5), out=out))
    t2 = t_ms(lambda: ops.linear_f16s(x, pk, ln=(g, b, 1e-5), post_residual=r, out=out))
    print(f'rows {rows} {cin}->{cout}: plain {t0 * 1e3:7.1f} us   +LN {t1 * 1e3:7.1f} us ({t1 / t0:.2f}x)   +LN+residual {t2 * 1e3:7.1f} us ({t2 / t0:.2f}x)')
