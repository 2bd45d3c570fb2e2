import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from far_amd import ops, _lib
lib = _lib.load()
g = torch.Generator(device='cuda').manual_seed(78)
D, H = 128, 8
ws = [torch.randn(D, D, device='cuda', generator=g) / 11 for _ in range(4)]
gam, bet = torch.rand(D, device='cuda', generator=g) + 0.5, torch.randn(D, device='cuda', generator=g) * 0.1
w0 = torch.randn(2 * D, 2 * D, device='cuda', generator=g) / 16
w2 = torch.randn(D, 2 * D, device='cuda', generator=g) / 16
pa, pm = ops.PackedAttn(*ws), ops.PackedMlp(w0, w2)
n = 30000
x = torch.randn(n, 25, D, device='cuda', generator=g)
s = torch.randn(n, 25, D, device='cuda', generator=g)
msg = ops.attn_block(x, s, pa, H, gam, bet, 1e-5)
W = torch.randn(1024, 35840, device='cuda', generator=g) / 190
pr = ops.PackedRows(W)
feats = torch.randn(8, 35840, device='cuda', generator=g)
ref = ops.rows_linear(feats, pr).clone()
from far_amd.loftr.transformer import LoFTREncoderLayer
torch.manual_seed(3)
layer = LoFTREncoderLayer(256, 8).cuda().eval()
xl = torch.randn(16, 4800, 256, device='cuda', generator=g)
xc = torch.randn(16, 120, 160, 128, device='cuda', generator=g).relu_()
wc = torch.randn(208, 128, 3, 3, device='cuda', generator=g) * 0.03
lnw = torch.ones(256, device='cuda'); lnb = torch.zeros(256, device='cuda')
f0 = torch.randn(8, 4800, 256, device='cuda', generator=g); f1 = torch.randn(8, 4800, 256, device='cuda', generator=g)
import far_amd.ops as O
names = [k for k in dir(O) if 'conv' in k.lower()]
print(names)
torch.cuda.synchronize()
side = torch.cuda.Stream()
def k9conv():
    pc = k9conv.pc
    return ops.conv_nhwc(xc, pc)
try:
    from far_amd.loftr.backbone import _PackCache
    import torch.nn as nn
    conv = nn.Conv2d(128, 208, 3, 2, 1, bias=False).cuda()
    k9conv.pc = _PackCache().get('c', conv, None, True)
except Exception as e:
    print('no k9 conv aggressor', e); k9conv = None
def k1():
    return ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8, variant='f16s')
aggr = {'K14 pipeline 0': lambda: ops.attn_block(x, s, pa, H, gam, bet, 1e-5),
        'K13 split': lambda: ops.mlp_fused(x, msg, pm, gam, bet, 1e-5),
        'K13 plain16': lambda: ops.mlp_fused(x, msg, pm, gam, bet, 1e-5, plain16=True),
        'K9 linear layer d256': lambda: layer(xl, xl),
        'K6 layernorm': lambda: ops.layernorm(xl, lnw, lnb, 1e-5),
        'K1 f16s': k1}
if k9conv: aggr['K9 conv 3x3/s2'] = k9conv
def sweep(tag):
    for name, fn in aggr.items():
        bad = 0
        with torch.no_grad():
            for it in range(10):
                for _ in range(3):
                    fn()
                with torch.cuda.stream(side):
                    y = ops.rows_linear(feats, pr)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                bad += not torch.equal(y, ref)
        print(f'{tag} victim K15 next to {name:22s}: {bad} of 10 launches differ')
sweep('')
for v in (1, 2):
    lib.far_set_tuning(11, v)
    aggr = {f'K14 pipeline {v}': aggr['K14 pipeline 0']}
    sweep('')
lib.far_set_tuning(11, 0)
