"""K2 parity on the GPU: far_emm_pv_f16s (split-fp16 operands, the default) and far_emm_pv_f32 (+ stats, exact-f32
MFMA) through the C ABI vs oracle/head.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(Z, N, seed, qk_amp=2.0):
    rng = np.random.default_rng(seed)
    q = (qk_amp * rng.standard_normal((Z, N, 64))).astype(np.float32)
    k = (qk_amp * rng.standard_normal((Z, N, 64))).astype(np.float32)
    v = rng.standard_normal((Z, N, 64)).astype(np.float32)
    pos = rng.uniform(-1, 1, (N, 6)).astype(np.float32)
    pos[:, 5] = 1
    return q, k, v, pos


@pytest.mark.parametrize('exact_f32', [False, True])
@pytest.mark.parametrize('Z,N', [(3, 192), (2, 221), (8, 64), (1, 4800)])
def test_emm_bilinear(Z, N, exact_f32):
    from far_amd import ops
    from oracle import head as oh
    q, k, v, pos = _inputs(Z, N, seed=N)
    F, T = ops.emm_bilinear(*(torch.from_numpy(a).cuda() for a in (q, k, v, pos)), 0.125, exact_f32=exact_f32)
    torch.cuda.synchronize()
    vt = np.concatenate([v, np.broadcast_to(pos, (Z, N, 6))], axis=2)
    Fref, A = oh.bilinear_attention(q, k, vt, 0.125, dtype=np.float64)
    Tref = A @ vt.astype(np.float64)
    np.testing.assert_allclose(T.cpu().numpy(), Tref, atol=2e-5 * np.abs(Tref).max(), rtol=1e-4)
    # regression-logit tolerance of north_star is 1e-3 relative; hold the 70x70 blocks much tighter
    np.testing.assert_allclose(F.cpu().numpy(), Fref, atol=1e-4 * np.abs(Fref).max(), rtol=1e-3)
    F32, _ = oh.bilinear_attention(q, k, vt, 0.125, dtype=np.float32)
    print('max|F - f64| / max|F| =', np.abs(F.cpu().numpy() - Fref).max() / np.abs(Fref).max(),
          ' fp32-restatement vs f64:', np.abs(F32 - Fref).max() / np.abs(Fref).max())


@pytest.mark.parametrize('Z,N', [(3, 192), (2, 221), (1, 4800)])
def test_emm_bilinear_plain_fp16_operands(Z, N):
    """far_emm_pv_f16 (round 5, LoFTR.set_precision('mixed16')): plain fp16 operands, fp32 accumulation and statistics.  Not the
    parity variant: the bar is the 16-bit-operand class (1e-3 relative, north_star's regression-logit tolerance) against the
    float64 oracle, and the kernel must be a different result from the split one (it really drops the low halves)."""
    from far_amd import ops
    from oracle import head as oh
    q, k, v, pos = _inputs(Z, N, seed=N)
    t = [torch.from_numpy(a).cuda() for a in (q, k, v, pos)]
    F, T = ops.emm_bilinear(*t, 0.125, plain16=True)
    Fs, Ts = ops.emm_bilinear(*t, 0.125)
    torch.cuda.synchronize()
    vt = np.concatenate([v, np.broadcast_to(pos, (Z, N, 6))], axis=2)
    Fref, A = oh.bilinear_attention(q, k, vt, 0.125, dtype=np.float64)
    Tref = A @ vt.astype(np.float64)
    eT = np.abs(T.cpu().numpy() - Tref).max() / np.abs(Tref).max()
    eF = np.abs(F.cpu().numpy() - Fref).max() / np.abs(Fref).max()
    eFs = np.abs(Fs.cpu().numpy() - Fref).max() / np.abs(Fref).max()
    print(f'plain fp16: max|T - f64|/max|T| = {eT:.3g}, max|F - f64|/max|F| = {eF:.3g} (split: {eFs:.3g})')
    assert torch.isfinite(T).all()
    assert eT < 3e-3 and eF < 3e-3
    assert eF > eFs                       # the split variant is the tighter one
    assert not torch.equal(T, Ts)


def test_emm_variants_agree_on_peaked_scores():
    """Large score range (|s| up to ~60): the one-exp formulation of the split variant must not under/overflow."""
    from far_amd import ops
    q, k, v, pos = _inputs(2, 320, seed=5, qk_amp=6.0)
    t = [torch.from_numpy(a).cuda() for a in (q, k, v, pos)]
    Fa, Ta = ops.emm_bilinear(*t, 0.125)
    Fb, Tb = ops.emm_bilinear(*t, 0.125, exact_f32=True)
    assert torch.isfinite(Ta).all()
    torch.testing.assert_close(Ta, Tb, atol=2e-5 * float(Tb.abs().max()), rtol=1e-4)
    torch.testing.assert_close(Fa, Fb, atol=1e-4 * float(Fb.abs().max()), rtol=1e-3)


def test_emm_planes_layout_matches_contiguous():
    """K2 reading q | k | v from the per-(tensor, head) planes of the fused projection == K2 on the permuted copies."""
    from far_amd import ops
    B, h, N = 2, 4, 256
    g = torch.Generator(device='cuda').manual_seed(9)
    planes = torch.randn(3 * h, 2 * B, N, 64, device='cuda', generator=g)
    pos = torch.rand(N, 6, device='cuda', generator=g)
    F, T = ops.emm_bilinear_planes(planes, pos, 0.125, B)
    q, k, v = (planes[t * h:(t + 1) * h].permute(1, 0, 2, 3) for t in range(3))       # (2B, h, N, 64): [image, pair]
    q = torch.cat([q[B:], q[:B]], 0)                                                   # direction d uses image 1 - d
    Fr, Tr = ops.emm_bilinear(*(t.reshape(2 * B * h, N, 64).contiguous() for t in (q, k, v)), pos, 0.125)
    assert torch.equal(T, Tr) and torch.equal(F, Fr)
