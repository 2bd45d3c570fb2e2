"""Diagnostic: K14 (fused attention block, d_model 128, windows) vs float64 and vs the separate launches; timing."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import ops

torch.manual_seed(0)
d, H = 128, 8
ws = [torch.randn(d, d, device='cuda') / d ** 0.5 for _ in range(4)]          # q, k, v, merge
g, b = torch.rand(d, device='cuda') + 0.5, torch.randn(d, device='cuda')
pa = ops.PackedAttn(*ws)
pcs = [ops.PackedConv(w) for w in ws]
rel = lambda a, r: float((a.double() - r.double()).abs().max() / r.double().abs().max())


def ref64(x, s):
    xd, sd = x.double(), s.double()
    q, k, v = xd @ ws[0].double().t(), sd @ ws[1].double().t(), sd @ ws[2].double().t()
    N, L, _ = q.shape
    S = k.shape[1]
    Q = torch.nn.functional.elu(q.view(N, L, H, 16)) + 1
    K = torch.nn.functional.elu(k.view(N, S, H, 16)) + 1
    V = v.view(N, S, H, 16) / S
    KV = torch.einsum('nshd,nshv->nhdv', K, V)
    Z = 1 / (torch.einsum('nlhd,nhd->nlh', Q, K.sum(1)) + 1e-6)
    msg = (torch.einsum('nlhd,nhdv,nlh->nlhv', Q, KV, Z) * S).reshape(N, L, d)
    return torch.nn.functional.layer_norm(msg @ ws[3].double().t(), (d,), g.double(), b.double(), 1e-5)


def separate(x, s):
    q = ops.linear_f16s(x, pcs[0])
    k = ops.linear_f16s(s, pcs[1])
    v = ops.linear_f16s(s, pcs[2])
    m = ops.linear_attention(q, k, v, H)
    return ops.linear_f16s(m, pcs[3], ln=(g, b, 1e-5))


for (N, L, S) in ((1, 25, 25), (3, 25, 25), (5, 25, 25), (700, 25, 25), (9, 32, 32), (6, 17, 9)):
    x, s = torch.randn(N, L, d, device='cuda'), torch.randn(N, S, d, device='cuda')
    y = ops.attn_block(x, s, pa, H, g, b, 1e-5)
    r = ref64(x, s)
    print(f'N={N} L={L} S={S}: fused vs float64 {rel(y, r):.2e}   separate launches vs float64 {rel(separate(x, s), r):.2e}')


def t_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


N = 61000
x, s = torch.randn(N, 25, d, device='cuda'), torch.randn(N, 25, d, device='cuda')
out = torch.empty_like(x)
print(f'fine-level shape N={N}: fused {t_ms(lambda: ops.attn_block(x, s, pa, H, g, b, 1e-5, out=out)) * 1e3:.1f} us   separate launches {t_ms(lambda: separate(x, s)) * 1e3:.1f} us')
