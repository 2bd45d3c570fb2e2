import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import ops
torch.manual_seed(0)
d = 128
w0 = torch.randn(2 * d, 2 * d, device='cuda') / (2 * d) ** 0.5
w2 = torch.randn(d, 2 * d, device='cuda') / (2 * d) ** 0.5
g, b = torch.rand(d, device='cuda') + 0.5, torch.randn(d, device='cuda')
pm = ops.PackedMlp(w0, w2)
p0, p2 = ops.PackedConv(w0), ops.PackedConv(w2)
R = 61000 * 25
x, m = torch.randn(1, R, d, device='cuda'), torch.randn(1, R, d, device='cuda')
out = torch.empty_like(x)
for _ in range(3):
    ops.mlp_fused(x, m, pm, g, b, 1e-5, out=out)
    hdn = ops.linear_f16s(x, p0, act='relu', x2=m)
    ops.linear_f16s(hdn, p2, ln=(g, b, 1e-5), post_residual=x, out=out)
torch.cuda.synchronize()
