"""Run the hand-written kernels in isolation at bench shapes (for rocprofv3 --pmc / --kernel-trace passes)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from far_amd import ops, _lib

which = sys.argv[1] if len(sys.argv) > 1 else 'all'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
it = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)
L = 4800
if which in ('k1', 'all'):
    f0 = 1.2 * torch.randn(n, L, 256, device=dev, generator=g)
    f1 = f0[:, torch.randperm(L, device=dev, generator=g)] + 0.1 * torch.randn(n, L, 256, device=dev, generator=g)
    for _ in range(it):
        ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0)
        ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, variant='f16s')
if which in ('k1conf', 'all'):
    f0 = 1.2 * torch.randn(n, L, 256, device=dev, generator=g)
    f1 = f0[:, torch.randperm(L, device=dev, generator=g)] + 0.1 * torch.randn(n, L, 256, device=dev, generator=g)
    for _ in range(it):
        ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, want_conf=True)
if which in ('k1w', 'all'):
    f0 = 1.2 * torch.randn(n, L, 256, device=dev, generator=g)
    f1 = f0[:, torch.randperm(L, device=dev, generator=g)] + 0.1 * torch.randn(n, L, 256, device=dev, generator=g)
    conf = torch.empty(n, L, L, device=dev)
    lib = _lib.load()
    for variant, tiles in ((0, 30), (0, 15), (0, 20)):     # wide-tile writer with 30 / 15 / 20 row tiles per work item
        lib.far_set_tuning(2, variant)
        lib.far_set_tuning(3, tiles)
        for _ in range(it):
            c, listed = ops.conf_matrix(f0, f1, 0.1, out=conf)
        ref = ops.coarse_match(f0[:2], f1[:2], 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, want_conf=True, variant='f16s')['conf_matrix']
        print('k1w variant', variant, tiles, ': listed', listed, 'entries =', listed / (n * L), 'per row; max dev vs fused writer',
              float((c[:2] - ref).abs().max()))
    lib.far_set_tuning(2, 0)
    lib.far_set_tuning(3, 0)
    del conf
if which in ('pmc',):
    # the kernels whose HBM traffic is reported: K9 196->196 3x3 @240x320 (the dominant kernel of the step), K9 linear,
    # the conf_matrix writer, K1's training backward
    x = torch.randn(2 * n, 240, 320, 196, device=dev, generator=g).relu_()
    w = torch.randn(196, 196, 3, 3, device=dev, generator=g) * 0.03
    pc = ops.PackedConv(w, torch.ones(196, device=dev), torch.zeros(196, device=dev))
    # K17, the kernel the inference step runs for this layer, in the layout the step uses: the 196-channel maps are stored with
    # 208 channels (zero weights for the extra ones; far_amd/loftr/backbone.py)
    wp = torch.zeros(208, 208, 3, 3, device=dev)
    wp[:196, :196] = w
    pw = ops.PackedWino(wp, torch.ones(208, device=dev), torch.zeros(208, device=dev))
    xp = torch.zeros(2 * n, 240, 320, 208, device=dev)
    xp[..., :196] = x
    for _ in range(it):
        ops.conv_nhwc(x, pc, act='relu')          # K9 (gradients enabled here: conv_nhwc does not dispatch to K17)
        ops.conv3x3_wino(xp, pw, act='relu')
    del x, xp
    r = torch.randn(1, 1, n * L, 256, device=dev, generator=g)
    pl = ops.PackedConv(torch.randn(256, 256, device=dev, generator=g) * 0.05)
    for _ in range(it):
        ops.conv_nhwc(r, pl)
    del r
    # round 4: the LinearAttention epilogues of K9 (k | v projection -> K'^T V state, q projection -> attention message) on the
    # tokens of 2n images, and FinePreprocess's merge_feat in gather mode on n x 1880 windows of a 240 x 320 fine map pair
    xs = torch.randn(2 * n, L, 256, device=dev, generator=g)
    wq, wk, wv = (torch.randn(256, 256, device=dev, generator=g) / 16 for _ in range(3))
    pq, pst = ops.PackedConv(wq), ops.PackedConv(ops.kv_interleaved_weight(wk, wv, 8))
    for _ in range(it):
        _, im = ops.linear_kv_state(xs, pst, L, want_image=True)
        ops.linear_q_apply(xs, pq, im, L)
    del xs
    fmap = torch.randn(2 * n, 240, 320, 128, device=dev, generator=g)
    M = 1880 * 2 * n
    bi = torch.sort(torch.randint(0, 2 * n, (M,), device=dev, generator=g))[0]
    ci = torch.randint(0, 60 * 80, (M,), device=dev, generator=g)
    pm = ops.PackedConv(torch.randn(128, 128, device=dev, generator=g) * 0.1)
    cw = torch.randn(M, 128, device=dev, generator=g)
    for _ in range(it):
        ops.linear_gather_f16s(fmap, bi, ci, 80, 5, 4, pm, residual=cw, res_group=25)
    del fmap
    f0 = 1.2 * torch.randn(n, L, 256, device=dev, generator=g)
    f1 = f0[:, torch.randperm(L, device=dev, generator=g)] + 0.1 * torch.randn(n, L, 256, device=dev, generator=g)
    conf = torch.empty(n, L, L, device=dev)
    for _ in range(it):
        ops.conf_matrix(f0, f1, 0.1, out=conf)
    del conf
    pb = torch.arange(4, device=dev).repeat_interleave(L)
    pi = torch.arange(L, device=dev).repeat(4)
    pj = torch.randint(0, L, (4 * L,), device=dev)
    a0, a1 = f0[:4].clone().requires_grad_(True), f1[:4].clone().requires_grad_(True)
    for _ in range(it):
        ops.coarse_pos_conf(a0, a1, pb, pi, pj, 0.1).sum().backward()
if which in ('util',):
    # matrix-pipe utilisation / clock / LDS passes: the split-precision kernels of the step at bench shapes + two
    # HBM-bound kernels as the clock reference
    for (cin, cout, hh, ww) in ((196, 196, 240, 320), (128, 128, 240, 320), (256, 256, 120, 160)):
        x = torch.randn(2 * n, hh, ww, cin, device=dev, generator=g).relu_()
        w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * 0.03
        pc = ops.PackedConv(w, torch.ones(cout, device=dev), torch.zeros(cout, device=dev))
        for _ in range(it):
            ops.conv_nhwc(x, pc, act='relu')
        del x
    r = torch.randn(1, 1, n * L, 512, device=dev, generator=g)
    pl = ops.PackedConv(torch.randn(512, 512, device=dev, generator=g) * 0.05)
    for _ in range(it):
        ops.conv_nhwc(r, pl)
    del r
    f0 = 1.2 * torch.randn(n, L, 256, device=dev, generator=g)
    f1 = f0[:, torch.randperm(L, device=dev, generator=g)] + 0.1 * torch.randn(n, L, 256, device=dev, generator=g)
    for _ in range(it):
        ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, variant='f16s')
    Z = n * 8
    q = torch.randn(Z, L, 64, device=dev, generator=g); k = torch.randn(Z, L, 64, device=dev, generator=g)
    v = torch.randn(Z, L, 64, device=dev, generator=g); pos = torch.rand(L, 6, device=dev, generator=g)
    for _ in range(it):
        ops.emm_bilinear(q, k, v, pos, 0.125)
    del q, k, v
    q = torch.randn(2 * n, L, 256, device=dev, generator=g); k = torch.randn(2 * n, L, 256, device=dev, generator=g)
    v = torch.randn(2 * n, L, 256, device=dev, generator=g)
    for _ in range(it):
        ops.linear_attention(q, k, v, 8)
    big = torch.empty(n, L, L, device=dev)
    for _ in range(it):
        big.fill_(1.0)
if which in ('k1b',):
    f0 = 1.2 * torch.randn(n, L, 256, device=dev, generator=g)
    f1 = f0[:, torch.randperm(L, device=dev, generator=g)] + 0.1 * torch.randn(n, L, 256, device=dev, generator=g)
    for _ in range(it):
        ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, bf16=True)
        ops.coarse_match(f0, f1, 0.1, 0.2, 2, (60, 80), (60, 80), 8.0, bf16=True, want_conf=True)
if which in ('k2', 'all'):
    Z = n * 8
    q = torch.randn(Z, L, 64, device=dev, generator=g); k = torch.randn(Z, L, 64, device=dev, generator=g)
    v = torch.randn(Z, L, 64, device=dev, generator=g); pos = torch.rand(L, 6, device=dev, generator=g)
    for _ in range(it):
        ops.emm_bilinear(q, k, v, pos, 0.125)
if which in ('k5', 'all'):
    q = torch.randn(2 * n, L, 256, device=dev, generator=g); k = torch.randn(2 * n, L, 256, device=dev, generator=g)
    v = torch.randn(2 * n, L, 256, device=dev, generator=g)
    for _ in range(it):
        ops.linear_attention(q, k, v, 8)
if which in ('k9', 'all'):
    x = torch.randn(2 * n, 240, 320, 196, device=dev, generator=g).relu_()
    w = torch.randn(196, 196, 3, 3, device=dev, generator=g) * 0.03
    pc = ops.PackedConv(w, torch.ones(196, device=dev), torch.zeros(196, device=dev))
    for _ in range(it):
        ops.conv_nhwc(x, pc, act='relu')
    del x
    r = torch.randn(1, 1, n * L, 256, device=dev, generator=g)
    pl = ops.PackedConv(torch.randn(256, 256, device=dev, generator=g) * 0.05)
    for _ in range(it):
        ops.conv_nhwc(r, pl)
torch.cuda.synchronize()
