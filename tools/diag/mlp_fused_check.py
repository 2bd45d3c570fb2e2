"""Diagnostic: K13 (fused MLP block, d_model 128) vs the two K9 launches and vs float64; timing at the fine-level shape."""
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
rel = lambda a, r: float((a.double() - r.double()).abs().max() / r.double().abs().max())


def two_launch(x, m):
    hdn = ops.linear_f16s(x, p0, act='relu', x2=m)
    return ops.linear_f16s(hdn, p2, ln=(g, b, 1e-5), post_residual=x)


for R in (1, 31, 256, 777 * 25, 1000):
    x, m = torch.randn(1, R, d, device='cuda'), torch.randn(1, R, d, device='cuda')
    y = ops.mlp_fused(x, m, pm, g, b, 1e-5)
    ref2 = two_launch(x, m)
    xd, md = x.double(), m.double()
    hd = torch.relu(torch.cat([xd, md], -1) @ w0.double().t()) @ w2.double().t()
    r64 = xd + torch.nn.functional.layer_norm(hd, (d,), g.double(), b.double(), 1e-5)
    print(f'R={R}: fused vs float64 {rel(y, r64):.2e}   two K9 launches vs float64 {rel(ref2, r64):.2e}   fused vs two launches {rel(y, ref2):.2e}')


def t_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


R = 61000 * 25
x, m = torch.randn(1, R, d, device='cuda'), torch.randn(1, R, d, device='cuda')
out = torch.empty_like(x)
print(f'fine-level shape R={R}: fused {t_ms(lambda: ops.mlp_fused(x, m, pm, g, b, 1e-5, out=out)) * 1e3:.1f} us   two K9 launches {t_ms(lambda: two_launch(x, m)) * 1e3:.1f} us')
