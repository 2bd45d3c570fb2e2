"""Diagnostic: K10 stem timing at the bench shape (64 images 480 x 640 -> 240 x 320 x 128)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import ops
x = torch.rand(64, 1, 480, 640, device='cuda')
w = torch.randn(128, 1, 7, 7, device='cuda') * 0.1
sc, sh = torch.rand(128, device='cuda') + 0.5, torch.randn(128, device='cuda')
for _ in range(3):
    ops.stem7x7(x, w, sc, sh)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.stem7x7(x, w, sc, sh)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'stem: {ms * 1e3:.1f} us  ({64 * 240 * 320 * 128 * 4 / ms / 1e9:.2f} TB/s written)')
