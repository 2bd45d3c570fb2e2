import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import ops
torch.manual_seed(0)
D, H = 128, 8
ws = [torch.randn(D, D, device='cuda') / D ** 0.5 for _ in range(4)]
gam, bet = torch.rand(D, device='cuda') + 0.5, torch.randn(D, device='cuda')
w0, w2 = torch.randn(2 * D, 2 * D, device='cuda') / 16, torch.randn(D, 2 * D, device='cuda') / 16
pa, pm = ops.PackedAttn(*ws), ops.PackedMlp(w0, w2)
x, s = torch.randn(20000, 25, D, device='cuda'), torch.randn(20000, 25, D, device='cuda')
a0 = ops.attn_block(x, s, pa, H, gam, bet, 1e-5)
m0 = ops.mlp_fused(x, s, pm, gam, bet, 1e-5)
for name, fn, ref in (('K14', lambda: ops.attn_block(x, s, pa, H, gam, bet, 1e-5), a0), ('K13', lambda: ops.mlp_fused(x, s, pm, gam, bet, 1e-5), m0)):
    bad = 0
    for it in range(30):
        y = fn()
        d = (y != ref)
        if d.any():
            bad += 1
            idx = d.nonzero()
            wins = idx[:, 0].unique()
            print(name, 'launch', it, 'differs:', int(d.sum()), 'elements in', len(wins), 'windows; first windows', wins[:8].tolist(),
                  'rows', idx[:, 1].unique()[:10].tolist(), 'channels', idx[:, 2].unique()[:12].tolist(), 'max |diff|', float((y - ref).abs().max()))
    print(name, 'bad launches:', bad, 'of 30')
