"""Achievable HBM write / copy bandwidth on this box (torch fill_ / copy_ of a conf_matrix-sized buffer):
the ceiling the K1 materialising kernel is priced against besides the 8 TB/s spec."""
import torch

n = 32 * 4800 * 4800
x = torch.empty(n, dtype=torch.float32, device='cuda')
y = torch.empty(n, dtype=torch.float32, device='cuda')


def t(fn, it=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / it


ms = t(lambda: x.fill_(1.0))
print(f'fill_ {n * 4 / 1e9:.2f} GB: {ms:.3f} ms = {n * 4 / ms / 1e6:.0f} GB/s write')
ms = t(lambda: y.copy_(x))
print(f'copy_ {n * 4 / 1e9:.2f} GB: {ms:.3f} ms = {2 * n * 4 / ms / 1e6:.0f} GB/s read+write')
ms = t(lambda: x.sum())
print(f'sum   {n * 4 / 1e9:.2f} GB: {ms:.3f} ms = {n * 4 / ms / 1e6:.0f} GB/s read')
