"""K1 parity on the GPU: far_coarse_match_f32 through the C ABI vs the oracle (oracle/coarse.py)."""
import numpy as np
import pytest
import torch

from tests.util import correlated_features

pytestmark = pytest.mark.gpu

ATOL64 = 1e-5
CFG = dict(thr=0.2, border_rm=2, dsmax_temperature=0.1)


def _run(f0, f1, hw, want_conf, border=2, thr=0.2):
    from far_amd import ops
    d = ops.coarse_match(torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), 0.1, thr, border,
                         hw, hw, 8.0, want_conf=want_conf)
    torch.cuda.synchronize()
    return d


def _check(f0, f1, hw, full_conf):
    from oracle import coarse as oc
    ref = oc.coarse_matching(f0, f1, CFG, hw, hw, (hw[0] * 8, hw[1] * 8))
    ref64 = oc.coarse_matching(f0, f1, CFG, hw, hw, (hw[0] * 8, hw[1] * 8), dtype=np.float64)
    got = _run(f0, f1, hw, want_conf=full_conf)
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
