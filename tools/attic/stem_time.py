"""K10 inference form: split-fp16 (default) against the exact-f32 instruction (FAR_TUNING=12=1), 64 images 480 x 640.
Usage: python tools/stem_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from far_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device='cuda').manual_seed(1)
img = torch.rand(64, 1, 480, 640, device='cuda', generator=g)
w = torch.randn(128, 1, 7, 7, device='cuda', generator=g) * 0.2
sc, sh = torch.rand(128, device='cuda', generator=g) + 0.5, torch.randn(128, device='cuda', generator=g) * 0.1
ref = torch.relu(torch.nn.functional.conv2d(img[:4].double(), w.double(), stride=2, padding=3) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).permute(0, 2, 3, 1)
for off in (0, 1, 0, 1):
    lib.far_set_tuning(12, off)
    y = ops.stem7x7(img, w, sc, sh)
    err = float((y[:4].double() - ref).abs().max() / ref.abs().max())
    t = min(bench.event_time_ms(lambda: ops.stem7x7(img, w, sc, sh), iters=10, warm=3) for _ in range(3))
    print(f'{"exact-f32" if off else "split-fp16"}: {t:.3f} ms   max error vs float64 {err:.2e} of max|ref|', flush=True)
lib.far_set_tuning(12, 0)
