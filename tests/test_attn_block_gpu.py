"""K14 (far_attn_block_f16s): q / k / v projections + linear attention + merge + norm1 of a LoFTR encoder layer at
d_model = 128 on <= 32-token sequences in one launch -- against float64 (transformer.py:51-61 + linear_attention.py:31-50
restated), against the separate launches, and the whole fused layer (K14 + K13) against the unfused layer."""
import pytest
import torch

pytestmark = pytest.mark.gpu
D, H = 128, 8


def _setup(seed, amp=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    ws = [torch.randn(D, D, device='cuda', generator=g) * (amp / D ** 0.5) for _ in range(4)]      # q, k, v, merge
    gam = torch.rand(D, device='cuda', generator=g) + 0.5
    bet = torch.randn(D, device='cuda', generator=g)
    return ws, gam, bet, g


def _ref64(x, s, ws, gam, bet):
    F = torch.nn.functional
    xd, sd = x.double(), s.double()
    q, k, v = xd @ ws[0].double().t(), sd @ ws[1].double().t(), sd @ ws[2].double().t()
    N, L, _ = q.shape
    S = k.shape[1]
    Q = F.elu(q.view(N, L, H, 16)) + 1                                              # linear_attention.py:31-32
    K = F.elu(k.view(N, S, H, 16)) + 1
    V = v.view(N, S, H, 16) / S                                                     # :43
    KV = torch.einsum('nshd,nshv->nhdv', K, V)                                      # :44
    Z = 1 / (torch.einsum('nlhd,nhd->nlh', Q, K.sum(1)) + 1e-6)                     # :45
    msg = (torch.einsum('nlhd,nhdv,nlh->nlhv', Q, KV, Z) * S).reshape(N, L, D)      # :46-50
    return F.layer_norm(msg @ ws[3].double().t(), (D,), gam.double(), bet.double(), 1e-5)


@pytest.mark.parametrize('N,L,S', [(1, 25, 25), (3, 25, 25), (5, 25, 25), (6, 25, 25), (777, 25, 25), (9, 32, 32), (6, 17, 9), (4, 1, 1)])
def test_attn_block_matches_float64_and_the_separate_launches(N, L, S):
    from far_amd import ops
    ws, gam, bet, g = _setup(N + L)
    x = torch.randn(N, L, D, device='cuda', generator=g)
    s = torch.randn(N, S, D, device='cuda', generator=g)
    y = ops.attn_block(x, s, ops.PackedAttn(*ws), H, gam, bet, 1e-5)
    ref = _ref64(x, s, ws, gam, bet)
    pcs = [ops.PackedConv(w) for w in ws]
    q, k, v = ops.linear_f16s(x, pcs[0]), ops.linear_f16s(s, pcs[1]), ops.linear_f16s(s, pcs[2])
    sep = ops.linear_f16s(ops.linear_attention(q, k, v, H), pcs[3], ln=(gam, bet, 1e-5))
    sc = float(ref.abs().max())
    e1, e2 = float((y.double() - ref).abs().max()) / sc, float((sep.double() - ref).abs().max()) / sc
    print(f'[k14] N={N} L={L} S={S}: fused vs float64 {e1:.2e}; separate launches vs float64 {e2:.2e}')
    assert e1 < 3e-6 and e2 < 3e-6                                   # measured 2.6e-7 .. 4.2e-7 for both
    assert torch.isfinite(y).all() and y.shape == x.shape


@pytest.mark.parametrize('N,L,S', [(3, 25, 25), (777, 25, 25), (6, 17, 9)])
def test_attn_block_plain_fp16_operands(N, L, S):
    """far_attn_block_f16 (LoFTR.set_precision('fp16')): the same block on plain fp16 operands.  Not the parity kernel: the bar is the
    16-bit-operand class (3e-3 of the output scale against float64), it must differ from the split result, and repeat bit for bit."""
    from far_amd import ops
    ws, gam, bet, g = _setup(N + L)
    x = torch.randn(N, L, D, device='cuda', generator=g)
    s = torch.randn(N, S, D, device='cuda', generator=g)
    pa = ops.PackedAttn(*ws)
    y = ops.attn_block(x, s, pa, H, gam, bet, 1e-5, plain16=True)
    ys = ops.attn_block(x, s, pa, H, gam, bet, 1e-5)
    ref = _ref64(x, s, ws, gam, bet)
    sc = float(ref.abs().max())
    e, es = float((y.double() - ref).abs().max()) / sc, float((ys.double() - ref).abs().max()) / sc
    print(f'[k14 plain] N={N} L={L} S={S}: plain vs float64 {e:.2e} (split {es:.2e})')
    assert torch.isfinite(y).all() and es < e < 3e-3
    assert torch.equal(y, ops.attn_block(x, s, pa, H, gam, bet, 1e-5, plain16=True))


def test_attn_block_self_attention_and_scales():
    """source = x (the 'self' layers), large / small weights and activations (power-of-two pre-scaling, elu on both sides)."""
    from far_amd import ops
    for amp, xamp in ((1.0, 1.0), (30.0, 1.0), (1e-2, 1.0), (1.0, 5.0), (1.0, 0.05)):
        ws, gam, bet, g = _setup(11, amp)
        x = torch.randn(50, 25, D, device='cuda', generator=g) * xamp
        y = ops.attn_block(x, x, ops.PackedAttn(*ws), H, gam, bet, 1e-5)
        ref = _ref64(x, x, ws, gam, bet)
        assert float((y.double() - ref).abs().max()) < 5e-6 * float(ref.abs().max()), (amp, xamp)


def test_attn_block_rejects_what_it_is_not_built_for():
    from far_amd import _lib, ops
    ws, gam, bet, g = _setup(2)
    pa = ops.PackedAttn(*ws)
    x = torch.randn(2, 40, D, device='cuda')
    with pytest.raises(_lib.FarHipError):
        ops.attn_block(x, x, pa, H, gam, bet, 1e-5)                   # 40 tokens: the generic path's job
    x = torch.randn(2, 25, D, device='cuda')
    with pytest.raises(_lib.FarHipError):
        ops.attn_block(x, x, pa, 4, gam, bet, 1e-5)                   # 4 heads of 32
    with pytest.raises(_lib.FarHipError):
        ops.attn_block(x, x, pa, H, gam, bet, 1e-5, out=x)            # out aliases the input
    assert ops.attn_block(x[:0], x[:0], pa, H, gam, bet, 1e-5).shape == (0, 25, D)


def test_fused_fine_layer_equals_the_unfused_layer():
    """LoFTREncoderLayer(128, 8) on fine-level-shaped windows: K14 + K13 (two launches) against the seven-launch path, self
    and cross; and the fine transformer (self + cross layers) end to end."""
    from far_amd.loftr.transformer import LocalFeatureTransformer, LoFTREncoderLayer
    torch.manual_seed(0)
    layer = LoFTREncoderLayer(128, 8).cuda().eval()
    for p in layer.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_uniform_(p)
    x = torch.randn(901, 25, 128, device='cuda')
    s = torch.randn(901, 25, 128, device='cuda')
    with torch.no_grad():
        for src in (x, s):
            LoFTREncoderLayer.fused_attn = LoFTREncoderLayer.fused_mlp = True
            a = layer(x, src)
            LoFTREncoderLayer.fused_attn = LoFTREncoderLayer.fused_mlp = False
            try:
                b = layer(x, src)
            finally:
                LoFTREncoderLayer.fused_attn = LoFTREncoderLayer.fused_mlp = True
            d = float((a - b).abs().max()) / float(b.abs().max())
            print(f'[fused fine layer] max relative difference {d:.2e}')
            assert d < 3e-6
        tr = LocalFeatureTransformer({'d_model': 128, 'd_ffn': 128, 'nhead': 8, 'layer_names': ['self', 'cross'], 'attention': 'linear'}).cuda().eval()
        a0, a1 = tr(x, s)
        LoFTREncoderLayer.fused_attn = LoFTREncoderLayer.fused_mlp = False
        try:
            b0, b1 = tr(x, s)
        finally:
            LoFTREncoderLayer.fused_attn = LoFTREncoderLayer.fused_mlp = True
        for a, b in ((a0, b0), (a1, b1)):
            assert float((a - b).abs().max()) < 5e-6 * float(b.abs().max())


def test_fused_fine_kernels_are_run_to_run_deterministic():
    """K13 / K14 synchronise their LDS weight ring by hand (counted vmcnt + raw barriers): a missing wait shows up as a
    run-to-run difference.  Fifty launches each on 20 k windows while a second stream keeps the memory system busy."""
    from far_amd import ops
    ws, gam, bet, g = _setup(77)
    w0 = torch.randn(2 * D, 2 * D, device='cuda', generator=g) / 16
    w2 = torch.randn(D, 2 * D, device='cuda', generator=g) / 16
    pa, pm = ops.PackedAttn(*ws), ops.PackedMlp(w0, w2)
    x = torch.randn(20000, 25, D, device='cuda', generator=g)
    s = torch.randn(20000, 25, D, device='cuda', generator=g)
    side = torch.cuda.Stream()
    junk = torch.empty(1 << 28, device='cuda')
    a0 = ops.attn_block(x, s, pa, H, gam, bet, 1e-5)
    m0 = ops.mlp_fused(x, a0, pm, gam, bet, 1e-5)
    for it in range(50):
        with torch.cuda.stream(side):
            junk.add_(1.0)
        a = ops.attn_block(x, s, pa, H, gam, bet, 1e-5)
        m = ops.mlp_fused(x, a, pm, gam, bet, 1e-5)
        assert torch.equal(a, a0), f'K14 differs at launch {it}'
        assert torch.equal(m, m0), f'K13 differs at launch {it}'
    side.synchronize()


def test_attn_block_overflow_is_reported():
    """K14 chains products of 2^4-scaled operands (K'^T V / S, then KV Q'): an input beyond 4094 -- or inputs of a few
    hundred whose per-head K'^T V sums pass 4094 -- gives inf / NaN and sets the device flag; in range it stays clear."""
    from far_amd import ops
    ws, gam, bet, g = _setup(31)
    x = torch.randn(40, 25, D, device='cuda', generator=g)
    ops.overflow_flag('cuda').zero_()
    y = ops.attn_block(x, x, ops.PackedAttn(*ws), H, gam, bet, 1e-5)
    assert torch.isfinite(y).all() and not ops.activation_overflowed('cuda')
    xb = x.clone()
    xb[7, 3, 9] = 9000.0
    y = ops.attn_block(xb, xb, ops.PackedAttn(*ws), H, gam, bet, 1e-5)
    assert not torch.isfinite(y[7]).all() and torch.isfinite(y[8:]).all()
    assert ops.activation_overflowed('cuda')
    y = ops.attn_block(x * 300.0, x * 300.0, ops.PackedAttn(*ws), H, gam, bet, 1e-5)     # |x| ~ 1000 < 4094, products are not
    print('[k14 product overflow] finite:', bool(torch.isfinite(y).all()), 'flag:', int(ops.overflow_flag('cuda').item()))
    assert bool(torch.isfinite(y).all()) != ops.activation_overflowed('cuda')           # never silently non-finite


@pytest.mark.parametrize('plain16', [False, True])
def test_fused_fine_kernels_are_deterministic_at_bench_scale(plain16):
    """The fine level of the bench step runs K14 / K13 on 120 296 windows (60 148 matches, 'self' layers on both images): at that
    size the round-3 form of K14 (two 4-wave workgroups per CU) differed run to run in a handful of windows once its elu got cheap
    (round 5; tools/fine_time.py).  Six launches each of the shipped kernels at that size, bit for bit, next to a busy second stream;
    plain16: the plain-fp16-operand forms of the 'fp16' mode (same pipeline, a third of the MFMAs)."""
    from far_amd import ops
    ws, gam, bet, g = _setup(78)
    w0 = torch.randn(2 * D, 2 * D, device='cuda', generator=g) / 16
    w2 = torch.randn(D, 2 * D, device='cuda', generator=g) / 16
    pa, pm = ops.PackedAttn(*ws), ops.PackedMlp(w0, w2)
    n = 120296
    x = torch.randn(n, 25, D, device='cuda', generator=g)
    s = torch.randn(n, 25, D, device='cuda', generator=g)
    side = torch.cuda.Stream()
    junk = torch.empty(1 << 27, device='cuda')
    a0 = ops.attn_block(x, s, pa, H, gam, bet, 1e-5, plain16=plain16)
    m0 = ops.mlp_fused(x, a0, pm, gam, bet, 1e-5, plain16=plain16)
    for it in range(6):
        with torch.cuda.stream(side):
            junk.add_(1.0)
        a = ops.attn_block(x, s, pa, H, gam, bet, 1e-5, plain16=plain16)
        m = ops.mlp_fused(x, a, pm, gam, bet, 1e-5, plain16=plain16)
        da = int(((a - a0).abs().flatten(1).max(1).values > 0).sum())
        dm = int(((m - m0).abs().flatten(1).max(1).values > 0).sum())
        assert da == 0, f'K14: {da} of {n} windows differ at launch {it}'
        assert dm == 0, f'K13: {dm} of {n} windows differ at launch {it}'
    side.synchronize()
