"""Diagnostic: does the K9 Linear launch pay for workgroup-count quantisation?  128-row workgroups, 2 per CU -> 512 resident;
153 600 rows (one image side of the 32-pair batch) = 1200 workgroups = 2.34 rounds."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import ops          # noqa: E402


def t_ms(fn, iters=20, warm=5):
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
for cin, cout in ((256, 256), (512, 512), (256, 768)):
    w = torch.randn(cout, cin, device='cuda') / cin ** 0.5
    pk = ops.PackedConv(w, split=True)
    for rows in (65536, 98304, 131072, 140000, 153600, 163840, 196608, 262144, 307200, 327680):
        x = torch.randn(1, rows, cin, device='cuda')
        out = torch.empty(1, rows, cout, device='cuda')
        ms = t_ms(lambda: ops.linear_f16s(x, pk, out=out))
        print(f'{cin}->{cout} rows {rows:7d} wgs {rows / 128:7.1f} rounds {rows / 128 / 512:5.2f}: {ms * 1e3:8.1f} us  '
              f'{ms * 1e3 / (rows / 128 / 512):7.1f} us/round-equivalent  {rows * (cin + cout) * 4 / ms / 1e9:6.2f} TB/s')
