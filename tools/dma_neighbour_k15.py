"""K15 (far_rows_linear_f32) on a second stream next to each of the library's kernels on the first: does the victim's result change?
Round 6 (docs/rounds/r06.md section 2f): with v_pk_fma_f32 in k_rows_partial it did, next to K13 / K14 / K9 -- 30 launches of 30, lanes
48..63 of the even accumulators -- and the build now compiles every file that can share a CU without the packed fp32 instructions
(far_amd/build.py: PACKED_FP32_FILES).  This script is the check on the shipped library: every line must say 0.
python tools/dma_neighbour_k15.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
big = torch.randn(32 << 20, device='cuda')
xc = torch.randn(64, 60, 80, 256, device='cuda', generator=g)
pw = ops.PackedWino(torch.randn(256, 256, 3, 3, device='cuda', generator=g) * 0.03)
torch.cuda.synchronize()
side = torch.cuda.Stream()
aggr = {'none': lambda: None,
        'K14': lambda: ops.attn_block(x, s, pa, H, gam, bet, 1e-5),
        'K13': lambda: ops.mlp_fused(x, msg, pm, gam, bet, 1e-5),
        'K17': lambda: ops.conv3x3_wino(xc, pw, act='relu'),
        'aten mul': lambda: big * 1.0001,
        'K15 itself': lambda: ops.rows_linear(feats, pr)}
for name, fn in aggr.items():
    bad = 0
    pat = set()
    for it in range(30):
        for _ in range(3):
            fn()
        with torch.cuda.stream(side):
            y = ops.rows_linear(feats, pr)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1
            df = y != ref
            pat |= {(int(r) % 2, int(c) % 64 // 16) for r, c in df.nonzero()[:2000].tolist()}
    print(f'victim K15 on a side stream next to {name:12s}: {bad} of 30 launches differ', sorted(pat)[:8])
# the reverse: K14 / K13 as victims next to K15
for name, fn, refv in (('K14', aggr['K14'], msg.clone()), ('K13', aggr['K13'], aggr['K13']().clone())):
    bad = 0
    for it in range(30):
        with torch.cuda.stream(side):
            for _ in range(4):
                ops.rows_linear(feats, pr)
        y = fn()
        torch.cuda.synchronize()
        bad += not torch.equal(y, refv)
    print(f'victim {name} next to K15 on a side stream: {bad} of 30 differ')
