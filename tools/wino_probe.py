"""K9 and K17 on one bench shape, a few launches each (for rocprofv3 --pmc / --kernel-trace passes: tools/wino_pmc.sh).
Usage: python tools/wino_probe.py [H W Cin Cout [N]]"""
import sys
sys.path.insert(0, '.')
import torch
from far_amd import ops

H, W, ci, co = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (240, 320, 128, 128)
N = int(sys.argv[5]) if len(sys.argv) >= 6 else 64
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(N, H, W, ci, device='cuda', generator=g).relu_()
w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * (2.0 / (ci * 9)) ** 0.5
pc = ops.PackedConv(w, torch.ones(co, device='cuda'), torch.zeros(co, device='cuda'))
pw = ops.PackedWino(w, torch.ones(co, device='cuda'), torch.zeros(co, device='cuda'))
for _ in range(4):
    ops.conv_nhwc(x, pc, act='relu')
    ops.conv3x3_wino(x, pw, act='relu')
torch.cuda.synchronize()
