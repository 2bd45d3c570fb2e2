"""K3 / K5 parity on the GPU through the C ABI vs oracle/fine.py, oracle/attention.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('channels_last', [False, True])
def test_fine_gather(channels_last):
    from far_amd import ops
    from oracle import fine as of
    rng = np.random.default_rng(0)
    N, C, Hf, Wf, wc = 2, 128, 48, 64, 16
    feat = rng.standard_normal((N, C, Hf, Wf)).astype(np.float32)
    M = 300
    b = np.sort(rng.integers(0, N, M)).astype(np.int64)
    cells = rng.integers(0, 12 * 16, M).astype(np.int64)
    cells[:4] = [0, 15, 11 * 16, 12 * 16 - 1]  # corners: zero padding on two sides
    t = torch.from_numpy(feat).cuda()
    if channels_last:
        t = t.contiguous(memory_format=torch.channels_last)
    got = ops.fine_gather(t, torch.from_numpy(b).cuda(), torch.from_numpy(cells).cuda(), wc, 5, 4)
    ref = of.unfold_windows(feat, b, cells, wc, 5, 4)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)  # a gather: bit exact
    # and against torch's own unfold on the device (the op the reference calls)
    unf = torch.nn.functional.unfold(torch.from_numpy(feat), (5, 5), stride=4, padding=2)
    unf = unf.view(N, C, 25, -1).permute(0, 3, 2, 1)[torch.from_numpy(b), torch.from_numpy(cells)]
    np.testing.assert_array_equal(got.cpu().numpy(), unf.numpy())


def test_fine_gather_empty():
    from far_amd import ops
    t = torch.zeros(1, 128, 8, 8, device='cuda')
    e = torch.zeros(0, dtype=torch.int64, device='cuda')
    assert ops.fine_gather(t, e, e, 2, 5, 4).shape == (0, 25, 128)


def test_fine_expect():
    from far_amd import ops
    from oracle import fine as of
    rng = np.random.default_rng(1)
    M, WW, C = 777, 25, 128
    f0 = rng.standard_normal((M, WW, C)).astype(np.float32)
    f1 = rng.standard_normal((M, WW, C)).astype(np.float32)
    f1[:50] = f0[:50, 12:13, :] * (rng.random((50, WW, 1)) > 0.8)  # peaked heatmaps
    mk = (rng.integers(0, 80, (M, 2)) * 8).astype(np.float32)
    expec, mk1 = ops.fine_expect(*(torch.from_numpy(a).cuda() for a in (f0, f1, mk)), 4.0)
    e64, m64 = of.fine_matching(f0, f1, mk, 4.0, dtype=np.float64)
    e32, m32 = of.fine_matching(f0, f1, mk, 4.0, dtype=np.float32)
    np.testing.assert_allclose(expec.cpu().numpy(), e64, atol=2e-5, rtol=0)
    np.testing.assert_allclose(mk1.cpu().numpy(), m64, atol=1e-4, rtol=0)
    np.testing.assert_allclose(expec.cpu().numpy(), e32, atol=1e-4, rtol=0)


@pytest.mark.parametrize('N,L,S,C', [(2, 4800, 4800, 256), (3, 100, 333, 256), (700, 25, 25, 128)])
def test_linear_attention(N, L, S, C):
    from far_amd import ops
    from oracle import attention as oa
    rng = np.random.default_rng(2)
    q = rng.standard_normal((N, L, C)).astype(np.float32)
    k = rng.standard_normal((N, S, C)).astype(np.float32)
    v = rng.standard_normal((N, S, C)).astype(np.float32)
    got = ops.linear_attention(*(torch.from_numpy(a).cuda() for a in (q, k, v)), 8).cpu().numpy()
    ref = oa.linear_attention(q, k, v, 8, dtype=np.float64)
    np.testing.assert_allclose(got, ref, atol=1e-5 * np.abs(ref).max(), rtol=1e-4)


@pytest.mark.parametrize('N,Hf,Wf,M,Cout', [(4, 48, 64, 300, 128), (2, 30, 44, 777, 128), (3, 24, 32, 1, 128), (2, 48, 64, 500, 256)])
def test_linear_reads_fine_windows_through_the_indices(N, Hf, Wf, M, Cout):
    """far_linear_gather_f16s (K9 gather mode): merge_feat's Linear layer on windows that are never stored == the same layer on
    fine_gather's window tensor, bit for bit (border windows with zero padding, grouped residual, ragged last tile)."""
    from far_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M + Hf)
    fmap = torch.randn(N, Hf, Wf, 128, device='cuda', generator=g)
    wc, hc = Wf // 4, Hf // 4
    b = torch.sort(torch.randint(0, N, (M,), device='cuda', generator=g))[0]
    cells = torch.randint(0, hc * wc, (M,), device='cuda', generator=g)
    cells[:min(M, 4)] = torch.tensor([0, wc - 1, (hc - 1) * wc, hc * wc - 1], device='cuda')[:min(M, 4)]     # the corners
    pc = ops.PackedConv(torch.randn(Cout, 128, device='cuda', generator=g) * 0.1, None, torch.randn(Cout, device='cuda', generator=g))
    res = torch.randn(M, Cout, device='cuda', generator=g)
    win = ops.fine_gather(fmap.permute(0, 3, 1, 2), b, cells, wc, 5, 4)
    ref = ops.linear_f16s(win, pc, residual=res, res_group=25)
    got = ops.linear_gather_f16s(fmap, b, cells, wc, 5, 4, pc, residual=res, res_group=25)
    assert got.shape == ref.shape and torch.equal(got, ref)
    assert torch.equal(ops.linear_gather_f16s(fmap, b, cells, wc, 5, 4, pc, act='relu'), ops.linear_f16s(win, pc, act='relu'))
    # a 1 x 1 "window": rows picked from a token tensor (down_proj on feat_c[b_ids, i_ids], fine_preprocess.py:50-51)
    tok = torch.randn(N, 1, hc * wc, 128, device='cuda', generator=g)
    assert torch.equal(ops.linear_gather_f16s(tok, b, cells, hc * wc, 1, 1, pc).view(M, -1), ops.linear_f16s(tok[b, 0, cells].contiguous(), pc))


@pytest.mark.parametrize('N,L,S', [(3, 4800, 4800), (2, 77, 100), (5, 200, 64), (2, 320, 6120), (4, 64, 150), (1, 33, 65), (3, 192, 128), (2, 6120, 6120), (3, 70, 130)])
def test_linear_kv_state_projection_fused_with_ktv(N, L, S):
    """far_linear_kv_f16s: the k | v projection of a d_model-256 layer ending in K'^T V (k, v never stored) + the apply half of K5.
    Against float64 (the state and the attention output), against the unfused launches, run to run, and image by image (the bits
    of an image's state must not depend on what else is in the launch, whatever the image length)."""
    from far_amd import ops
    from oracle import attention as oa
    rng = np.random.default_rng(S + L)
    src = rng.standard_normal((N, S, 256)).astype(np.float32)
    xq = rng.standard_normal((N, L, 256)).astype(np.float32)
    wq, wk, wv = (rng.standard_normal((256, 256)).astype(np.float32) / 16 for _ in range(3))
    cu = lambda a: torch.from_numpy(a).cuda()
    pkv = ops.PackedConv(ops.kv_interleaved_weight(cu(wk), cu(wv), 8), split=True)
    kv = ops.linear_kv_state(cu(src), pkv, S)
    k64, v64 = src.astype(np.float64) @ wk.T.astype(np.float64), src.astype(np.float64) @ wv.T.astype(np.float64)
    kf = (np.where(k64 > 0, k64, np.expm1(k64)) + 1).reshape(N, S, 8, 32)
    ref_kv = np.einsum('nshd,nshv->nhdv', kf, v64.reshape(N, S, 8, 32) / S).reshape(N, 256, 32)
    ref_ks = kf.sum(1).reshape(N, 256)
    got = kv.cpu().numpy()
    np.testing.assert_allclose(got[:, :, :32], ref_kv, atol=3e-6 * np.abs(ref_kv).max(), rtol=0)
    np.testing.assert_allclose(got[:, :, 32], ref_ks, atol=3e-6 * np.abs(ref_ks).max(), rtol=0)
    q = ops.linear_f16s(cu(xq), ops.PackedConv(cu(wq), split=True))
    out = ops.linear_attention_apply(q, kv, 8, S)
    ref = oa.linear_attention(xq.astype(np.float64) @ wq.T.astype(np.float64), k64, v64, 8, dtype=np.float64)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-5 * np.abs(ref).max(), rtol=1e-4)
    k, v = ops.linear_f16s(cu(src), ops.PackedConv(torch.cat([cu(wk), cu(wv)], 0), split=True), out_planes=2)
    unf = ops.linear_attention(q, k, v, 8)
    d = float((out - unf).abs().max()) / float(unf.abs().max())
    print(f'[kv state] N={N} L={L} S={S}: fused vs unfused max rel diff {d:.2e}')
    assert d < 2e-6
    assert torch.equal(kv, ops.linear_kv_state(cu(src), pkv, S))
    if L >= 64:              # the q projection with the apply half in its epilogue (far_linear_q_apply_f16s): q never stored
        kv2, image = ops.linear_kv_state(cu(src), pkv, S, want_image=True)
        assert torch.equal(kv2, kv)
        pq = ops.PackedConv(cu(wq), split=True)
        msg = ops.linear_q_apply(cu(xq), pq, image, S)
        np.testing.assert_allclose(msg.cpu().numpy(), ref, atol=1e-5 * np.abs(ref).max(), rtol=1e-4)
        d2 = float((msg - unf).abs().max()) / float(unf.abs().max())
        print(f'[q apply] N={N} L={L} S={S}: fused vs unfused max rel diff {d2:.2e}')
        assert d2 < 2e-6
        assert torch.equal(msg, ops.linear_q_apply(cu(xq), pq, image, S))
        one = ops.linear_q_apply(cu(xq[N - 1:]), pq, image[-image.numel() // N:], S)
        assert torch.equal(one, msg[N - 1:])
    for n in range(N):      # an image's 64-row blocks are its own (padded launch geometry when S % 64 != 0): the same sums in the same
        assert torch.equal(kv[n:n + 1], ops.linear_kv_state(cu(src[n:n + 1]), pkv, S)), f'image {n}'      # order whatever else is in the launch


@pytest.mark.parametrize('N,L,S', [(3, 4800, 4800), (2, 77, 100), (3, 70, 130)])
def test_linear_kv_state_and_q_apply_on_plain_fp16_operands(N, L, S):
    """The same two launches with plain fp16 operands in the projections' K loops (PackedConv(split=False);
    LoFTR.set_precision('fp16')): 16-bit-operand bar against float64, repeatable, images independent of the batch."""
    from far_amd import ops
    from oracle import attention as oa
    rng = np.random.default_rng(S + L)
    src = rng.standard_normal((N, S, 256)).astype(np.float32)
    xq = rng.standard_normal((N, L, 256)).astype(np.float32)
    wq, wk, wv = (rng.standard_normal((256, 256)).astype(np.float32) / 16 for _ in range(3))
    cu = lambda a: torch.from_numpy(a).cuda()
    pkv = ops.PackedConv(ops.kv_interleaved_weight(cu(wk), cu(wv), 8), split=False)
    pq = ops.PackedConv(cu(wq), split=False)
    kv, image = ops.linear_kv_state(cu(src), pkv, S, want_image=True)
    msg = ops.linear_q_apply(cu(xq), pq, image, S)
    k64, v64 = src.astype(np.float64) @ wk.T.astype(np.float64), src.astype(np.float64) @ wv.T.astype(np.float64)
    ref = oa.linear_attention(xq.astype(np.float64) @ wq.T.astype(np.float64), k64, v64, 8, dtype=np.float64)
    e = np.abs(msg.cpu().numpy() - ref).max() / np.abs(ref).max()
    print(f'[kv state + q apply, plain fp16] N={N} L={L} S={S}: vs float64 {e:.2e}')
    assert torch.isfinite(msg).all() and 1e-6 < e < 3e-3
    assert torch.equal(msg, ops.linear_q_apply(cu(xq), pq, image, S))
    kv1, im1 = ops.linear_kv_state(cu(src[N - 1:]), pkv, S, want_image=True)
    assert torch.equal(kv1, kv[N - 1:])
    assert torch.equal(ops.linear_q_apply(cu(xq[N - 1:]), pq, im1, S), msg[N - 1:])


@pytest.mark.parametrize('N,L,S', [(700, 25, 25), (5, 32, 32), (7, 17, 9), (3, 1, 1), (4, 9, 30)])
def test_linear_attention_short_windows_fused_kernel(N, L, S):
    """The one-kernel form used for short sequences with 16-channel heads (the fine-level windows): against the float64
    oracle, and against the generic two-kernel path (same products in the same order: equal to the last bit or two)."""
    from far_amd import _lib, ops
    from oracle import attention as oa
    rng = np.random.default_rng(L * 100 + S)
    q = rng.standard_normal((N, L, 128)).astype(np.float32)
    k = rng.standard_normal((N, S, 128)).astype(np.float32)
    v = rng.standard_normal((N, S, 128)).astype(np.float32)
    t = [torch.from_numpy(a).cuda() for a in (q, k, v)]
    got = ops.linear_attention(*t, 8)
    ref = oa.linear_attention(q, k, v, 8, dtype=np.float64)
    np.testing.assert_allclose(got.cpu().numpy(), ref, atol=1e-5 * np.abs(ref).max(), rtol=1e-4)
    lib = _lib.load()
    lib.far_set_tuning(4, 1)                     # A/B knob: generic path
    try:
        gen = ops.linear_attention(*t, 8)
    finally:
        lib.far_set_tuning(4, 0)
    d = float((got - gen).abs().max()) / float(gen.abs().max())
    print(f'[k5 window] N={N} L={L} S={S}: fused vs generic max rel diff {d:.2e}')
    assert d < 1e-6


@pytest.mark.parametrize('rows,C,res', [(4800 * 3, 256, True), (1000, 256, False), (777 * 25, 128, True), (33, 512, False), (10, 200, True)])
def test_layernorm(rows, C, res):
    from far_amd import ops
    rng = np.random.default_rng(C + rows)
    x = (3 * rng.standard_normal((rows, C)) + 0.5).astype(np.float32)
    g = rng.uniform(0.5, 1.5, C).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    r = rng.standard_normal((rows, C)).astype(np.float32) if res else None
    cu = lambda a: None if a is None else torch.from_numpy(a).cuda()
    got = ops.layernorm(cu(x), cu(g), cu(b), 1e-5, cu(r)).cpu().numpy()
    x64 = x.astype(np.float64)
    ref = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * g + b
    if res:
        ref = ref + r
    np.testing.assert_allclose(got, ref, atol=2e-6 * np.abs(ref).max(), rtol=1e-5)


@pytest.mark.parametrize('fmt', [torch.contiguous_format, torch.channels_last])
def test_backbone_epilogues_match_torch(fmt):
    """K7 (folded BN + residual + activation) and K8 (2x bilinear upsample + add) vs the torch ops they replace."""
    from far_amd import ops
    import torch.nn.functional as F
    g = torch.Generator(device='cuda').manual_seed(0)
    N, C, H, W = 3, 196, 30, 40
    x = torch.randn(N, C, H, W, device='cuda', generator=g).contiguous(memory_format=fmt)
    r = torch.randn(N, C, H, W, device='cuda', generator=g).contiguous(memory_format=fmt)
    bn = torch.nn.BatchNorm2d(C).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        for act, fn in [('relu', F.relu), ('none', lambda t: t), ('leaky', lambda t: F.leaky_relu(t, 0.01))]:
            for res in [None, r]:
                ref = fn(bn(x) + (res if res is not None else 0))
                got = ops.affine_act(x, scale, shift, residual=res, act=act, inplace=False)
                assert got.is_contiguous(memory_format=fmt)
                torch.testing.assert_close(got, ref, atol=2e-6, rtol=1e-5)
        lo = torch.randn(N, C, 15, 20, device='cuda', generator=g).contiguous(memory_format=fmt)
        ref = x + F.interpolate(lo, scale_factor=2., mode='bilinear', align_corners=True)
        torch.testing.assert_close(ops.upsample2x_add(lo, x), ref, atol=2e-6, rtol=1e-5)
