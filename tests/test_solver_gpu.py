"""K4 parity on the GPU: far_solver_f64 through the C ABI vs oracle/solver.py (float64, bit-exact masks)."""
import numpy as np
import pytest
import torch

from tests.util import two_view_scene

pytestmark = pytest.mark.gpu

H = 512


def _run(scenes, mode, priors=None, pcl=None, seed=11, H=H, samples=None, thresh=0.5):
    from far_amd import ops
    k0 = np.concatenate([s[0] for s in scenes])
    k1 = np.concatenate([s[1] for s in scenes])
    offs = np.concatenate([[0], np.cumsum([len(s[0]) for s in scenes])]).astype(np.int32)
    K = np.stack([s[2] for s in scenes])
    many = mode != 'ransac'
    thr = np.array([3e-7 if many else (thresh / np.mean([k[0, 0], k[1, 1], k[0, 0], k[1, 1]])) ** 2 for k in K])
    cu = lambda a, dt=None: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = ops.solve_pose_batch(cu(k0), cu(k1), offs, cu(K), cu(K), cu(thr), many,
                               priorRT=cu(priors.astype(np.float32)) if priors is not None else None,
                               pcl=cu(pcl), H=H, seed=seed, samples=cu(samples), debug=True)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}, offs


def _compare(scenes, mode, priors=None, pcl=None, seed=11):
    from oracle import solver as osv
    got, offs = _run(scenes, mode, priors, pcl, seed)
    for b, (k0, k1, K, Rgt, tgt) in enumerate(scenes):
        pr = None if priors is None else priors[b]
        solver = {'ransac': 'ransac', 'prior': 'prior_ransac', 'noprior': 'prior_ransac_noprior'}[mode]
        ret, nafter, tight, ultra, dbg = osv.estimate_pose(k0, k1, K, K, 0.5, solver=solver, priorRT=pr, seed=seed,
                                                            pair=b, H=H, pcl=pcl)
        np.testing.assert_array_equal(got['samples'][b], dbg['samples'])          # integer path: bit exact
        v = dbg['valid']
        # well-conditioned hypotheses agree to float64 round-off; judge on the Sampson counts, which is what
        # the selection consumes, and on F for the winning model
        same = got['count_all'][b][v] == dbg['count'][v]
        assert same.mean() > 0.995, same.mean()
        assert np.all(np.isinf(got['score_all'][b][~v]))
        assert got['best'][b] == dbg['best']
        np.testing.assert_allclose(got['E'][b], dbg['F'][dbg['best']], rtol=1e-7, atol=1e-9)
        sl = slice(offs[b], offs[b + 1])
        assert (ret is not None) == bool(got['status'][b])
        if ret is not None:
            R, t, mask, E = ret
            np.testing.assert_array_equal(got['mask'][sl].astype(bool), mask)      # inlier mask: bit exact
            assert np.linalg.norm(got['R'][b] - R) < 1e-8 and np.linalg.norm(got['t'][b] - t) < 1e-8
            # and it is a sane pose
            assert np.linalg.norm(R - Rgt) < 0.05, np.linalg.norm(R - Rgt)
            assert min(np.linalg.norm(t - tgt), np.linalg.norm(t + tgt)) < 0.15
        assert got['num_after'][b] == nafter and got['tight'][b] == tight and got['ultra'][b] == ultra


def test_ransac_branch():
    scenes = [two_view_scene(M, seed=s) for s, M in enumerate([300, 1000, 64, 2000])]
    _compare(scenes, 'ransac')


def test_noprior_branch():
    scenes = [two_view_scene(M, seed=10 + s, outlier_frac=0.5) for s, M in enumerate([500, 800])]
    _compare(scenes, 'noprior')


def test_prior_branch():
    rng = np.random.default_rng(0)                                     # metrics.py:103 (np.random.seed(0) upstream)
    pcl = rng.uniform(-3.0, 3.0, (300, 3)).astype(np.float32)
    scenes = [two_view_scene(M, seed=20 + s, outlier_frac=0.4) for s, M in enumerate([400, 1200, 900])]
    priors = []
    for (_, _, _, R, t) in scenes:
        d = 0.05 * rng.standard_normal(3)
        Rn = R @ (np.eye(3) + np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]]))
        priors.append(np.concatenate([Rn, (2.5 * t + 0.05 * rng.standard_normal(3))[:, None]], 1))
    _compare(scenes, 'prior', np.stack(priors), pcl)


def test_degenerate_inputs():
    # fewer than 8 correspondences and an empty pair: status 0, nothing crashes (metrics.py:83-85 analogue)
    s_ok = two_view_scene(200, seed=3)
    s_few = tuple(a[:6] if i < 2 else a for i, a in enumerate(two_view_scene(50, seed=4)))
    s_none = tuple(a[:0] if i < 2 else a for i, a in enumerate(two_view_scene(50, seed=5)))
    got, offs = _run([s_few, s_ok, s_none], 'ransac')
    assert list(got['status']) == [0, 1, 0]
    assert got['num_after'][0] == 0 and got['num_after'][2] == 0
    assert got['mask'][offs[0]:offs[1]].sum() == 0
    # a whole batch without a single correspondence (two blank pairs early in training): legal, every status 0
    pri = np.stack([np.eye(3, 4)] * 2)
    pcl = np.random.RandomState(0).uniform(-3, 3, (300, 3)).astype(np.float32)
    for mode in ('ransac', 'prior'):
        got, offs = _run([s_none, s_none], mode, pri if mode == 'prior' else None, pcl if mode == 'prior' else None)
        assert list(got['status']) == [0, 0] and list(got['num_after']) == [0, 0] and len(got['mask']) == 0


def test_explicit_samples_match_reference_style_call():
    # same hypothesis index sets fed to both sides (the protocol for parity with run_8point)
    from oracle import solver as osv
    sc = two_view_scene(700, seed=33, outlier_frac=0.2)
    rng = np.random.default_rng(1)
    samples = np.stack([rng.choice(len(sc[0]), 8, replace=False) for _ in range(H)]).astype(np.int32)
    got, _ = _run([sc], 'noprior', samples=samples[None])
    kn0, kn1 = osv.normalize_keypoints(sc[0], sc[1], sc[2], sc[2])
    kp1 = kn0.astype(np.float32).astype(np.float64)
    kp2 = kn1.astype(np.float32).astype(np.float64)
    F = osv.run_8point(kp1[samples], kp2[samples])
    err = np.abs(got['F_all'][0] - F).reshape(H, -1).max(1) / np.abs(F).reshape(H, -1).max(1)
    assert np.median(err) < 1e-10 and (err < 1e-6).mean() > 0.97, (np.median(err), (err < 1e-6).mean())


def test_cached_path_batch_256():
    """BASELINE configs[3]: solver on cached correspondences, batch 256, M <= 2000, H = 2048 (one launch set)."""
    from oracle import solver as osv
    rng = np.random.default_rng(7)
    B = 256
    Ms = rng.integers(100, 2000, B)
    scenes = [two_view_scene(int(M), seed=1000 + b, outlier_frac=0.3) for b, M in enumerate(Ms)]
    got, offs = _run(scenes, 'noprior', H=2048)
    assert got['status'].mean() > 0.97
    for b in [0, 37, 128, 255]:
        k0, k1, K, Rgt, tgt = scenes[b]
        ret, nafter, tight, ultra, dbg = osv.estimate_pose(k0, k1, K, K, 0.5, solver='prior_ransac_noprior', seed=11,
                                                            pair=b, H=2048)
        assert got['best'][b] == dbg['best']
        np.testing.assert_array_equal(got['mask'][offs[b]:offs[b + 1]].astype(bool), ret[2])
        assert np.linalg.norm(got['R'][b] - ret[0]) < 1e-8 and got['num_after'][b] == nafter
    # accuracy over the batch against ground truth
    Rerr = np.array([np.linalg.norm(got['R'][b] - scenes[b][3]) for b in range(B) if got['status'][b]])
    assert np.median(Rerr) < 0.02


def test_estimate_pose_single_pair_contract():
    """far_amd.solver.estimate_pose keeps the reference's signature and return contract (metrics.py:80-174, :169)."""
    from far_amd.solver import estimate_pose
    k0, k1, K, Rgt, tgt = two_view_scene(500, seed=77)
    Kt = torch.from_numpy(K).cuda()
    ret, n_after, tight, ultra = estimate_pose(torch.from_numpy(k0).cuda(), torch.from_numpy(k1).cuda(), Kt, Kt, 0.5,
                                               conf=0.99999, solver='prior_ransac', priorRT=None)
    R, t, mask, E = ret
    assert R.is_cuda and R.dtype == torch.float64 and R.shape == (3, 3) and t.shape == (3,) and t.is_cuda
    assert isinstance(mask, np.ndarray) and mask.dtype == bool and mask.shape == (500,)
    assert (not E.is_cuda) and E.dtype == torch.float64 and E.shape == (3, 3)
    assert int(n_after) == int(mask.sum()) and tight == 0 and ultra == 0       # plain-RANSAC branch (metrics.py:96-97)
    assert np.linalg.norm(R.cpu().numpy() - Rgt) < 0.05
    prior = np.concatenate([Rgt, (2 * tgt)[:, None]], 1)
    ret2, n2, tight2, ultra2 = estimate_pose(torch.from_numpy(k0).cuda(), torch.from_numpy(k1).cuda(), Kt, Kt, 0.5,
                                             solver='prior_ransac', priorRT=prior,
                                             translation_scale=torch.tensor(2.5, dtype=torch.float64))
    assert ret2 is not None and tight2 >= ultra2 > 0
    assert abs(float(ret2[1].norm()) - 2.5) < 1e-9                              # t *= translation_scale (:167-168)
    # fewer than 5 correspondences -> (None, 0, 0, 0) without touching the GPU (:83-85)
    assert estimate_pose(torch.from_numpy(k0[:4]).cuda(), torch.from_numpy(k1[:4]).cuda(), Kt, Kt, 0.5) == (None, 0, 0, 0)


def test_spvs_rt_is_order_independent_like_the_reference():
    """supervision.py:209-210 selects a pair's correspondences with `m_bids == bs`; in training m_bids is unsorted
    for B > 1 (coarse_matching.py:216-240).  The batched call site must give every pair ITS correspondences whatever
    the order, return the inlier mask in the caller's order, and ignore a stale `match_counts`."""
    from far_amd.config import RunCfg
    from far_amd.supervision import compute_supervision_RT
    scenes = [two_view_scene(M, seed=40 + s) for s, M in enumerate([400, 250, 600])]
    k0 = np.concatenate([s[0] for s in scenes])
    k1 = np.concatenate([s[1] for s in scenes])
    bid = np.concatenate([np.full(len(s[0]), b) for b, s in enumerate(scenes)])
    K = torch.from_numpy(np.stack([s[2] for s in scenes])).cuda()
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def run(perm, extra=None):
        b = cu(bid[perm])
        d = {'mkpts0_f': cu(k0[perm]), 'mkpts1_f': cu(k1[perm]), 'm_bids': b, 'b_ids': b, 'K0': K, 'K1': K.clone()}
        d.update(extra or {})
        compute_supervision_RT(d, RunCfg('prior_ransac_noprior'), H=H, seed=5)
        return d

    ident = np.arange(len(bid))
    ref = run(ident, {'match_counts': torch.tensor([len(s[0]) for s in scenes])})
    # interleave the pairs round-robin (the order INSIDE a pair is kept: sample indices address positions in a pair)
    pos = [list(np.nonzero(bid == b)[0]) for b in range(3)]
    rr = []
    while any(pos):
        for p in pos:
            if p:
                rr.append(p.pop(0))
    rr = np.array(rr)
    got = run(rr, {'match_counts': torch.tensor([1, 2, 3])})          # stale counts must be ignored, not trusted
    np.testing.assert_array_equal(got['loftr_rt'].cpu().numpy(), ref['loftr_rt'].cpu().numpy())
    for k in ('num_correspondences', 'num_correspondences_before_ransac', 'inliers_best_tight'):
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k].cpu().numpy())
    np.testing.assert_array_equal(got['solver_inlier_mask'].cpu().numpy(), ref['solver_inlier_mask'].cpu().numpy()[rr])
    assert np.linalg.norm(ref['loftr_rt'].cpu().numpy()[1, :, :3] - scenes[1][3]) < 0.05
    with pytest.raises(ValueError):
        bad = {'mkpts0_f': cu(k0), 'mkpts1_f': cu(k1[:-1]), 'm_bids': cu(bid), 'K0': K, 'K1': K}
        compute_supervision_RT(bad, RunCfg('ransac'), H=64)


@pytest.mark.parametrize('tag,mode', [('p', 'prior'), ('n', 'noprior')])
def test_kernel_vs_reference_whole_ransac_loop(tag, mode):
    """far_solver_f64 on the committed sample indices against the reference's own RANSAC.forward (golden G12,
    ransac.py:340-442): same selected hypothesis, same inlier sets off the decision margin (bars and measured
    agreement in tests/test_oracle_golden.py:g12_expectations)."""
    import os
    from tests.test_oracle_golden import g12_expectations
    from oracle import solver as osv
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g12_ransac_loop.npz'))
    scene = (g[f'{tag}_kpts0'], g[f'{tag}_kpts1'], g[f'{tag}_K'], g[f'{tag}_R_gt'], g[f'{tag}_t_gt'])
    priors = g['p_prior'][None] if tag == 'p' else None
    pcl = g['p_pcl'] if tag == 'p' else None
    smp = g[f'{tag}_samples'].astype(np.int32)[None]
    got, offs = _run([scene], mode, priors, pcl, H=smp.shape[1], samples=smp)
    valid = np.isfinite(got['score_all'][0])
    wq = None
    if tag == 'p':      # the kernel's integer weights are internal; the oracle's are held to the golden on CPU
        wq = osv.estimate_pose(scene[0], scene[1], scene[2], scene[2], 0.5, solver='prior_ransac', priorRT=priors[0],
                               pcl=pcl, samples=smp[0])[4]['wq']
    g12_expectations(tag, valid, got['count_all'][0], np.where(valid, got['score_all'][0], -np.inf), int(got['best'][0]),
                     wq, (int(got['num_after'][0]), int(got['tight'][0]), int(got['ultra'][0])),
                     got['mask'].astype(bool), g)
    assert np.linalg.norm(got['R'][0] - g[f'{tag}_R_gt']) < 0.03
