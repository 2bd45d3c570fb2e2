"""K1 parity on the GPU: far_coarse_match_f32 through the C ABI vs the oracle (oracle/coarse.py)."""
import numpy as np
import pytest
import torch

from tests.util import correlated_features

pytestmark = pytest.mark.gpu

ATOL64 = 1e-5
CFG = dict(thr=0.2, border_rm=2, dsmax_temperature=0.1)


def _run(f0, f1, hw, want_conf, border=2, thr=0.2, variant='f32'):
    from far_amd import ops
    d = ops.coarse_match(torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), 0.1, thr, border,
                         hw, hw, 8.0, want_conf=want_conf, variant=variant)
    torch.cuda.synchronize()
    return d


def _check(f0, f1, hw, full_conf, variant='f32'):
    from oracle import coarse as oc
    ref = oc.coarse_matching(f0, f1, CFG, hw, hw, (hw[0] * 8, hw[1] * 8))
    ref64 = oc.coarse_matching(f0, f1, CFG, hw, hw, (hw[0] * 8, hw[1] * 8), dtype=np.float64)
    got = _run(f0, f1, hw, want_conf=full_conf, variant=variant)
    # indices are only well defined away from the discontinuities: require the margin on the oracle side
    rg, cg, tg = oc.margins(ref['conf_matrix'], CFG['thr'])
    sel = ref['conf_matrix'].max(axis=2) > 0.05
    assert rg[sel].min() > 1e-5 and tg[sel].min() > 1e-5, 'test input lacks margin'
    for k in ['b_ids', 'i_ids', 'j_ids']:
        assert got[k].dtype == torch.int64
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])
    # tolerance vs the fp32 restatement = the fp32 reference's own deviation from exact arithmetic
    # (softmax swamping, see oracle/coarse.py); vs the float64 evaluation the kernel is held much tighter.
    np.testing.assert_allclose(got['mconf'].cpu().numpy(), ref['mconf'], atol=2e-4, rtol=0)
    np.testing.assert_array_equal(ref64['i_ids'], ref['i_ids'])
    np.testing.assert_allclose(got['mconf'].cpu().numpy(), ref64['mconf'], atol=ATOL64, rtol=0)
    np.testing.assert_array_equal(got['mkpts0_c'].cpu().numpy(), ref['mkpts0_c'])
    np.testing.assert_array_equal(got['mkpts1_c'].cpu().numpy(), ref['mkpts1_c'])
    if full_conf:
        np.testing.assert_allclose(got['conf_matrix'].cpu().numpy(), ref['conf_matrix'], atol=2e-4, rtol=0)
        err = np.abs(got['conf_matrix'].cpu().numpy() - ref64['conf_matrix']).max()
        print('max |conf - float64 oracle| =', err)
        assert err <= ATOL64
    return len(ref['b_ids'])


def test_small_full_conf():
    f0, f1, _ = correlated_features(2, (12, 16), 64, seed=1, amp=2.0)
    M = _check(f0, f1, (12, 16), True)
    assert M > 50


def test_ragged_tile_edges():
    # L = S = 13*17 = 221: not a multiple of the 128/32 tiles
    f0, f1, _ = correlated_features(3, (13, 17), 32, seed=2, amp=3.0)
    M = _check(f0, f1, (13, 17), True)
    assert M > 30


def test_full_grid_640x480():
    f0, f1, _ = correlated_features(1, (60, 80), 256, seed=3, amp=1.2, frac=0.8)
    M = _check(f0, f1, (60, 80), True)
    assert M > 2000


def test_no_matches():
    rng = np.random.default_rng(5)
    f0 = rng.standard_normal((1, 192, 64)).astype(np.float32) * 0.1
    f1 = rng.standard_normal((1, 192, 64)).astype(np.float32) * 0.1
    got = _run(f0, f1, (12, 16), False)
    assert got['b_ids'].numel() == 0 and got['mkpts0_c'].shape == (0, 2)


def test_permutation_recovered_batch():
    # size-independent property at full size: with f1 = f0[perm] (+ small noise) every interior cell must
    # match its permuted partner
    N, hw = 4, (60, 80)
    f0, f1, perms = correlated_features(N, hw, 256, seed=7, amp=1.5, noise=0.05)
    got = _run(f0, f1, hw, False)
    b, i, j = (got[k].cpu().numpy() for k in ['b_ids', 'i_ids', 'j_ids'])
    assert np.all(np.diff(b * 4800 + i) > 0), 'not ordered by (b, i)'
    for n in range(N):
        inv = np.argsort(perms[n])  # f1[j] = f0[perm[j]]  ->  partner of i is inv[i]
        sel = b == n
        np.testing.assert_array_equal(j[sel], inv[i[sel]])

        def interior(idx):
            y, x = idx // 80, idx % 80
            return (y >= 2) & (y < 58) & (x >= 2) & (x < 78)
        expect = np.nonzero(interior(np.arange(4800)) & interior(inv))[0]
        np.testing.assert_array_equal(i[sel], expect)


def test_padding_masks_and_image_scales():
    """mask_c0/mask_c1 (-1e9 fill, coarse_matching.py:110-113), mask_border_with_padding (:28-43) and
    scale0/scale1 (:247-254): exercised against a torch restatement of those reference lines."""
    from far_amd import ops
    rng = np.random.default_rng(9)
    N, hw, C = 2, (12, 16), 64
    f0, f1, _ = correlated_features(N, hw, C, seed=4, amp=2.0)
    L = hw[0] * hw[1]
    m0 = np.zeros((N, hw[0], hw[1]), bool)
    m1 = np.zeros((N, hw[0], hw[1]), bool)
    ext = [(10, 13, 12, 16), (12, 16, 9, 14)]
    for n, (h0, w0, h1, w1) in enumerate(ext):
        m0[n, :h0, :w0] = True
        m1[n, :h1, :w1] = True
    sc0 = rng.uniform(0.8, 1.3, (N, 2)).astype(np.float32)
    sc1 = rng.uniform(0.8, 1.3, (N, 2)).astype(np.float32)
    # reference arithmetic in float64
    a = torch.from_numpy(f0).double() / C ** .5
    b = torch.from_numpy(f1).double() / C ** .5
    sim = torch.einsum('nlc,nsc->nls', a, b) / 0.1
    valid = torch.from_numpy(m0.reshape(N, L))[..., None] & torch.from_numpy(m1.reshape(N, L))[:, None]
    sim = sim.masked_fill(~valid, -1e9)
    conf = sim.softmax(1) * sim.softmax(2)
    mask = (conf > 0.2).reshape(N, *hw, *hw).clone()
    bd = 2
    mask[:, :bd] = False; mask[:, :, :bd] = False; mask[:, :, :, :bd] = False; mask[:, :, :, :, :bd] = False
    for n, (h0, w0, h1, w1) in enumerate(ext):
        mask[n, h0 - bd:] = False; mask[n, :, w0 - bd:] = False
        mask[n, :, :, h1 - bd:] = False; mask[n, :, :, :, w1 - bd:] = False
    mask = mask.reshape(N, L, L) & (conf == conf.max(2, keepdim=True)[0]) & (conf == conf.max(1, keepdim=True)[0])
    mv, aj = mask.max(2)
    bi, ii = torch.where(mv)
    jj = aj[bi, ii]
    vh = torch.tensor(ext, dtype=torch.int32).cuda()
    got = ops.coarse_match(torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), 0.1, 0.2, bd, hw, hw, 8.0,
                           mask0=torch.from_numpy(m0.reshape(N, L).astype(np.uint8)).cuda(),
                           mask1=torch.from_numpy(m1.reshape(N, L).astype(np.uint8)).cuda(), valid_hw=vh,
                           scale0=torch.from_numpy(sc0).cuda(), scale1=torch.from_numpy(sc1).cuda())
    assert len(bi) > 20
    np.testing.assert_array_equal(got['b_ids'].cpu().numpy(), bi.numpy())
    np.testing.assert_array_equal(got['i_ids'].cpu().numpy(), ii.numpy())
    np.testing.assert_array_equal(got['j_ids'].cpu().numpy(), jj.numpy())
    mk0 = torch.stack([ii % hw[1], ii // hw[1]], 1) * (8.0 * torch.from_numpy(sc0)[bi])
    mk1 = torch.stack([jj % hw[1], jj // hw[1]], 1) * (8.0 * torch.from_numpy(sc1)[bi])
    np.testing.assert_allclose(got['mkpts0_c'].cpu().numpy(), mk0.numpy(), rtol=1e-6)
    np.testing.assert_allclose(got['mkpts1_c'].cpu().numpy(), mk1.numpy(), rtol=1e-6)


def test_mapfree_grid_544x720():
    # BASELINE configs[4]: 544x720 images -> 68x90 coarse grid, L = S = 6120 (not a multiple of the 128 tile)
    f0, f1, _ = correlated_features(2, (68, 90), 256, seed=11, amp=1.3, frac=0.9)
    M = _check(f0, f1, (68, 90), False)
    assert M > 6000


def _bf16_round(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize('N,hw,C', [(2, (12, 16), 256), (1, (60, 80), 256), (2, (13, 17), 256)])
def test_bf16_variant_is_exact_on_bf16_rounded_inputs(N, hw, C):
    """far_coarse_match_bf16: the contraction runs on the bf16 matrix core.  Its parity statement is against the
    oracle evaluated on the SAME bf16-rounded features (bit-exact ids on rows with margin, 1e-5 confidences)."""
    from far_amd import ops
    from oracle import coarse as oc
    f0, f1, _ = correlated_features(N, hw, C, seed=21, amp=1.3 if C == 256 else 2.0, frac=0.85)
    r0, r1 = _bf16_round(f0), _bf16_round(f1)
    ref = oc.coarse_matching(r0, r1, CFG, hw, hw, (hw[0] * 8, hw[1] * 8), dtype=np.float64)
    got = ops.coarse_match(torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), 0.1, 0.2, 2, hw, hw, 8.0,
                           want_conf=True, bf16=True)
    rg, cg, tg = oc.margins(ref['conf_matrix'], 0.2)
    sel = ref['conf_matrix'].max(axis=2) > 0.05
    assert rg[sel].min() > 1e-5 and tg[sel].min() > 1e-5
    for k in ['b_ids', 'i_ids', 'j_ids']:
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])
    np.testing.assert_allclose(got['mconf'].cpu().numpy(), ref['mconf'], atol=1e-5, rtol=0)
    np.testing.assert_allclose(got['conf_matrix'].cpu().numpy(), ref['conf_matrix'], atol=1e-5, rtol=0)
    assert len(ref['b_ids']) > 50


def test_bf16_variant_vs_fp32_path_iou():
    """Deviation of the bf16 variant from the exact fp32 path (documented, not parity): match-set IoU and the
    confidence difference on the common matches."""
    from far_amd import ops
    f0, f1, _ = correlated_features(4, (60, 80), 256, seed=23, amp=1.2, frac=0.8)
    a = ops.coarse_match(torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), 0.1, 0.2, 2, (60, 80), (60, 80), 8.0)
    b = ops.coarse_match(torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), 0.1, 0.2, 2, (60, 80), (60, 80), 8.0,
                         bf16=True)
    sa = set(zip(a['b_ids'].tolist(), a['i_ids'].tolist(), a['j_ids'].tolist()))
    sb = set(zip(b['b_ids'].tolist(), b['i_ids'].tolist(), b['j_ids'].tolist()))
    iou = len(sa & sb) / len(sa | sb)
    print('bf16 vs fp32 match-set IoU', iou, len(sa), len(sb))
    assert iou > 0.98


# ---- split-fp16 variant (far_coarse_match_f16s, C = 256): the same parity bar as the exact-f32 kernels -------------

@pytest.mark.parametrize('N,hw,seed,amp', [(2, (12, 16), 21, 1.6), (3, (13, 17), 22, 1.6), (1, (60, 80), 3, 1.2)])
def test_f16s_variant_parity(N, hw, seed, amp):
    f0, f1, _ = correlated_features(N, hw, 256, seed=seed, amp=amp, frac=0.8)
    M = _check(f0, f1, hw, True, variant='f16s')
    assert M > 30


def test_f16s_variant_mapfree_grid_and_agreement_with_f32():
    f0, f1, _ = correlated_features(2, (68, 90), 256, seed=11, amp=1.3, frac=0.9)
    M = _check(f0, f1, (68, 90), False, variant='f16s')
    assert M > 6000
    a, b = _run(f0, f1, (68, 90), False, variant='f16s'), _run(f0, f1, (68, 90), False)
    for k in ('b_ids', 'i_ids', 'j_ids'):
        assert torch.equal(a[k], b[k])
    assert float((a["mconf"] - b["mconf"]).abs().max()) < 1e-5          # both are within 1e-5 of the float64 oracle


def test_f16s_variant_no_matches_and_masks():
    from far_amd import ops
    rng = np.random.default_rng(5)
    f0 = rng.standard_normal((1, 192, 256)).astype(np.float32) * 0.1
    f1 = rng.standard_normal((1, 192, 256)).astype(np.float32) * 0.1
    got = _run(f0, f1, (12, 16), False, variant='f16s')
    assert got['b_ids'].numel() == 0 and got['mkpts0_c'].shape == (0, 2)
    # padded masks + valid extents + scales: must equal the exact-f32 kernels decision for decision
    N, hw = 2, (12, 16)
    f0, f1, _ = correlated_features(N, hw, 256, seed=4, amp=1.6)
    L = hw[0] * hw[1]
    m0 = np.zeros((N, hw[0], hw[1]), np.uint8); m1 = np.zeros((N, hw[0], hw[1]), np.uint8)
    ext = [(10, 13, 12, 16), (12, 16, 9, 14)]
    for n, (h0, w0, h1, w1) in enumerate(ext):
        m0[n, :h0, :w0] = 1; m1[n, :h1, :w1] = 1
    kw = dict(mask0=torch.from_numpy(m0.reshape(N, L)).cuda(), mask1=torch.from_numpy(m1.reshape(N, L)).cuda(),
              valid_hw=torch.tensor(ext, dtype=torch.int32).cuda(),
              scale0=torch.tensor([[1.1, 0.9], [1.0, 1.2]]).cuda(), scale1=torch.tensor([[0.8, 1.0], [1.3, 1.1]]).cuda())
    t0, t1 = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
    a = ops.coarse_match(t0, t1, 0.1, 0.2, 2, hw, hw, 8.0, variant='f16s', want_conf=True, **kw)
    b = ops.coarse_match(t0, t1, 0.1, 0.2, 2, hw, hw, 8.0, variant='f32', want_conf=True, **kw)
    assert a['b_ids'].numel() > 20
    for k in ('b_ids', 'i_ids', 'j_ids', 'mkpts0_c', 'mkpts1_c'):
        assert torch.equal(a[k], b[k]), k
    assert float((a['conf_matrix'] - b['conf_matrix']).abs().max()) < 1e-5


@pytest.mark.parametrize('N,hw,seed,amp,frac', [(1, (60, 80), 3, 1.2, 0.8), (2, (68, 90), 11, 1.3, 0.9), (8, (60, 80), 23, 1.2, 0.8)])
def test_conf_matrix_writer_is_fp32_grade(N, hw, seed, amp, frac):
    """far_conf_matrix_f16s (the HBM-bound writer: plain-fp16 scores + exact recomputation of every entry above 2^-12)
    against the float64 oracle: <= 1e-5 everywhere, like the split-precision matcher's writer; ragged sizes included
    (68x90: L = S = 6120 is not a multiple of the 64-column tile)."""
    from far_amd import ops
    from oracle import coarse as oc
    f0, f1, _ = correlated_features(N, hw, 256, seed=seed, amp=amp, frac=frac)
    t0, t1 = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
    conf, listed = ops.conf_matrix(t0, t1, 0.1)
    ref_fused = ops.coarse_match(t0, t1, 0.1, 0.2, 2, hw, hw, 8.0, want_conf=True, variant='f16s')['conf_matrix']
    L = hw[0] * hw[1]
    assert conf.shape == (N, L, L) and 0 < listed < N * L * 4
    worst = 0.0
    for n in range(min(N, 2)):          # the float64 oracle of one 4800 x 4800 matrix takes a few seconds
        ref64 = oc.conf_matrix(f0[n:n + 1], f1[n:n + 1], 0.1, dtype=np.float64)[0]
        worst = max(worst, float(np.abs(conf[n].cpu().numpy() - ref64).max()))
    dev_fused = float((conf - ref_fused).abs().max())
    big = ref_fused > 2.0 ** -11
    print(f'[conf writer] N={N} hw={hw}: listed {listed} entries ({listed / (N * L):.2f} per row), '
          f'max|conf - float64| = {worst:.2e}, max|conf - fused split writer| = {dev_fused:.2e}, '
          f'on entries > 2^-11: {float((conf - ref_fused).abs()[big].max()):.2e}')
    assert worst <= ATOL64 and dev_fused <= 2e-5


def test_conf_matrix_writer_with_masks():
    """Padded masks (coarse_matching.py:110-113: masked pairs get -1e9 before the softmaxes)."""
    from far_amd import ops
    from oracle import coarse as oc
    N, hw = 2, (24, 32)
    f0, f1, _ = correlated_features(N, hw, 256, seed=31, amp=1.4)
    L = hw[0] * hw[1]
    m0 = np.ones((N, L), np.uint8)
    m1 = np.ones((N, L), np.uint8)
    m0[0, 500:] = 0
    m1[1, 600:] = 0
    conf, _ = ops.conf_matrix(torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), 0.1,
                              torch.from_numpy(m0).cuda(), torch.from_numpy(m1).cuda())
    ref64 = oc.conf_matrix(f0, f1, 0.1, m0.astype(bool), m1.astype(bool), dtype=np.float64)
    valid = m0[..., None].astype(bool) & m1[:, None].astype(bool)
    got = conf.cpu().numpy()
    assert np.abs(got - ref64)[valid].max() <= ATOL64
    assert got[~valid].max() <= 1e-30 or np.abs(got - ref64)[~valid].max() <= ATOL64


@pytest.mark.parametrize('kind', ['peaky', 'diffuse', 'mixed_norms', 'near_threshold'])
def test_f16s_match_pass_prescreen_changes_nothing(kind):
    """Round 4: k1_match skips the tiles for which k1_screen's hi.hi scores prove that no entry can exceed `thr` (bound in the kernel).  With the prescreen on (default) and off (tuning key 10) the matches and their confidences must be
    IDENTICAL bit for bit -- on confident permutation matches, on a diffuse matrix where nothing may be skipped, with rows whose
    norms differ by 30x (the bound uses the largest norm), and with confidences straddling the threshold."""
    from far_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    hw = (40, 48)
    L = hw[0] * hw[1]
    if kind == 'peaky':
        f0, f1, _ = correlated_features(2, hw, 256, seed=31, amp=1.2, frac=0.9)
    elif kind == 'diffuse':
        f0 = (0.25 * rng.standard_normal((2, L, 256))).astype(np.float32)
        f1 = (0.25 * rng.standard_normal((2, L, 256))).astype(np.float32)
    elif kind == 'mixed_norms':
        f0, f1, _ = correlated_features(2, hw, 256, seed=32, amp=1.0, frac=0.8)
        sc = np.where(rng.random((2, L, 1)) < 0.1, 6.0, 0.2).astype(np.float32)
        f0, f1 = f0 * sc, f1 * sc
    else:
        f0, f1, _ = correlated_features(2, hw, 256, seed=33, amp=0.88, noise=0.1, frac=0.9)       # best confidences around 0.2
    outs = []
    for off in (0, 1, 0):
        lib.far_set_tuning(10, off)
        try:
            outs.append(_run(f0, f1, hw, False, variant='f16s'))
        finally:
            lib.far_set_tuning(10, 0)
    a, b, c = outs
    print(f'[prescreen] {kind}: {len(a["b_ids"])} matches')
    for k in ('b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts0_c', 'mkpts1_c'):
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    if kind == 'peaky':
        assert len(a['b_ids']) > 2000
    if kind == 'near_threshold':
        m = a['mconf']
        assert len(m) > 20 and float(m.min()) < 0.3            # matches close to the threshold exist
