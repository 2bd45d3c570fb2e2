"""The FPN merge 1x1 convolution (resnet_fpn.py:108-109, 113-114: lateral 1x1 + 2x bilinear upsample of the coarser level) at the bench
size, with and without the fused upsample residual: what the `up` epilogue costs.  Usage: python tools/fpn_merge_time.py"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from far_amd import ops

g = torch.Generator(device='cuda').manual_seed(1)
for (N, H, W, Cin, Cout) in ((64, 240, 320, 128, 208), (64, 120, 160, 196, 256), (64, 120, 160, 208, 256)):
    x = torch.randn(N, H, W, Cin, device='cuda', generator=g)
    up = torch.randn(N, H // 2, W // 2, Cout, device='cuda', generator=g)
    w = torch.randn(Cout, Cin, 1, 1, device='cuda', generator=g) * (1.0 / Cin) ** 0.5
    pc = ops.PackedConv(w)
    out = torch.empty(N, H, W, Cout, device='cuda')
    t0 = min(bench.event_time_ms(lambda: ops.conv_nhwc(x, pc, out=out), iters=5, warm=2) for _ in range(3))
    t1 = min(bench.event_time_ms(lambda: ops.conv_nhwc(x, pc, up=up, out=out), iters=5, warm=2) for _ in range(3))
    gb = (x.numel() + out.numel()) * 4 / 1e9
    fl = 2.0 * N * H * W * Cin * Cout * 3
    print(f'{Cin}->{Cout} @{H}x{W}: plain {t0:.3f} ms, with up {t1:.3f} ms; floors: HBM {gb / 5.4:.3f} (+ up {up.numel() * 4 / 1e9 / 5.4:.3f}) ms, MFMA {fl / 1.65e12:.3f} ms')
