"""Diagnostic: training-mode pieces at the fine-level shape (960 windows x 25 tokens x 128 channels)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import ops, autograd_ops as ag                   # noqa: E402
from far_amd.loftr.transformer import LoFTREncoderLayer      # noqa: E402

rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))
torch.manual_seed(0)
for (rows, cin, cout) in [(960 * 25, 128, 128), (960 * 25, 256, 256), (960 * 25, 256, 128), (4800, 256, 256), (24000, 256, 256)]:
    w = (torch.randn(cout, cin, device='cuda') / cin ** 0.5).requires_grad_(True)
    x = torch.randn(960 if rows == 24000 else 1, rows // (960 if rows == 24000 else 1), cin, device='cuda').requires_grad_(True)
    g = torch.randn(*x.shape[:2], cout, device='cuda')
    pk = ops.PackCache()
    y = ops.linear_train(x, w, None, pk, ('t', 0), split=True)
    y.backward(g)
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = None
    yr = torch.nn.functional.linear(x.double(), w.double())
    yr.backward(g.double())
    print(f'linear_train rows={rows} {cin}->{cout} shape {tuple(x.shape)}: y {rel(y, yr):.2e} dx {rel(gx, x.grad):.2e} dw {rel(gw, w.grad):.2e}')
    x.grad = w.grad = None

for (N, L, H, D) in [(3, 25, 8, 16), (960, 25, 8, 16), (960, 25, 8, 32)]:
    q, k, v, g = (torch.randn(N, L, H * D, device='cuda') for _ in range(4))
    a = [t.clone().requires_grad_(True) for t in (q, k, v)]
    out = ops.linear_attention_train(a[0], a[1], a[2], H, None, None)
    out.backward(g)
    r = [t.double().clone().requires_grad_(True) for t in (q, k, v)]
    ref = ag.linear_attention(r[0], r[1], r[2], H, None, None)
    ref.backward(g.double())
    print(f'k5 N={N} L={L} D={D}: out {rel(out, ref):.2e} ' + ' '.join(f'd{n} {rel(x.grad, y.grad):.2e}' for n, x, y in zip('qkv', a, r)))

for d in (128, 256):
    layer = LoFTREncoderLayer(d, 8).cuda().train()
    for p in layer.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_uniform_(p)
    x0, s0, g = (torch.randn(960, 25, d, device='cuda') for _ in range(3))
    res = {}
    for mode in (True, False):
        layer.hip_training = mode
        layer.zero_grad()
        x, s = x0.clone().requires_grad_(True), s0.clone().requires_grad_(True)
        y = layer(x, s)
        y.backward(g)
        res[mode] = (y.detach(), x.grad, s.grad, {k: p.grad.clone() for k, p in layer.named_parameters()})
    print(f'layer d={d}: y {rel(res[True][0], res[False][0]):.2e} dx {rel(res[True][1], res[False][1]):.2e} ds {rel(res[True][2], res[False][2]):.2e}')
    for k in res[False][3]:
        print(f'   d{k}: {rel(res[True][3][k], res[False][3][k]):.2e}')
