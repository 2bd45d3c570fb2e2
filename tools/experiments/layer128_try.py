import sys, torch, numpy as np
sys.path.insert(0, '.')
import bench
from far_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
d = 128
W = lambda o, i: torch.randn(o, i, device='cuda', generator=g) * (1.0 / i) ** 0.5
wq, wk, wv, wm, w0, w2 = W(d, d), W(d, d), W(d, d), W(d, d), W(2 * d, 2 * d), W(d, 2 * d)
n1 = (torch.rand(d, device='cuda', generator=g) + 0.5, torch.randn(d, device='cuda', generator=g), 1e-5)
n2 = (torch.rand(d, device='cuda', generator=g) + 0.5, torch.randn(d, device='cuda', generator=g), 1e-5)
pa, pm, pl = ops.PackedAttn(wq, wk, wv, wm), ops.PackedMlp(w0, w2), ops.PackedLayer128(wq, wk, wv, wm, w0, w2)
for N in (7, 64, 256, 600, 1001, 4000, 61000):
    x = torch.randn(N, 25, d, device='cuda', generator=g)
    src = torch.randn(N, 25, d, device='cuda', generator=g)
    msg = ops.attn_block(x, src, pa, 8, n1[0], n1[1], n1[2])
    ref = ops.mlp_fused(x, msg, pm, n2[0], n2[1], n2[2])
    got = ops.layer128_fused(x, src, pl, 8, n1, n2)
    dmax = float((got - ref).abs().max()) / float(ref.abs().max())
    print(N, 'fused vs two launches max rel diff', dmax, 'repeat equal', bool(torch.equal(got, ops.layer128_fused(x, src, pl, 8, n1, n2))))
    if N == 61000:
        t2 = min(bench.event_time_ms(lambda: ops.mlp_fused(x, ops.attn_block(x, src, pa, 8, n1[0], n1[1], n1[2]), pm, n2[0], n2[1], n2[2]), iters=5, warm=2) for _ in range(3))
        t1 = min(bench.event_time_ms(lambda: ops.layer128_fused(x, src, pl, 8, n1, n2), iters=5, warm=2) for _ in range(3))
        print('two launches %.3f ms, fused %.3f ms' % (t2, t1))
