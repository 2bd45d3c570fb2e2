"""Does running the backbone's layers on small image chunks (so that a layer's input is still in the 256 MiB die-level cache when
the next layer reads it) help K17 / K9?  Chain of four 128->128 3x3 layers @240x320 over 64 images: all images per layer, or
chunks of n images through all four layers.  Usage: python tools/chunk_ab.py"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from far_amd import ops

g = torch.Generator(device='cuda').manual_seed(1)
NI, H, W, C = 64, 240, 320, 128
x = torch.randn(NI, H, W, C, device='cuda', generator=g).relu_()
ws = [torch.randn(C, C, 3, 3, device='cuda', generator=g) * (2.0 / (C * 9)) ** 0.5 for _ in range(4)]
one, zero = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
pws = [ops.PackedWino(w, one, zero) for w in ws]
pcs = [ops.PackedConv(w, one, zero) for w in ws]
bufs = [torch.empty_like(x) for _ in range(2)]
ops.USE_WINO = False


def run(n, wino):
    for i in range(0, NI, n):
        src = x[i:i + n]
        for l in range(4):
            dst = bufs[l & 1][i:i + n]
            if wino:
                ops.conv3x3_wino(src, pws[l], act='relu', out=dst)
            else:
                ops.conv_nhwc(src, pcs[l], act='relu', out=dst)
            src = dst


for wino in (True, False):
    for n in (64, 16, 8, 4, 2, 1):
        t = min(bench.event_time_ms(lambda: run(n, wino), iters=3, warm=1) for _ in range(2))
        print(f'{"K17" if wino else "K9 "} chunks of {n:2d} images: {t:.3f} ms for 4 layers x 64 images ({t / 4:.3f} per layer)', flush=True)
