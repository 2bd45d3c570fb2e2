"""K4 parity on the GPU: far_solver_f64 through the C ABI vs oracle/solver.py (float64, bit-exact masks)."""
import numpy as np
import pytest
import torch

from tests.util import two_view_scene

pytestmark = pytest.mark.gpu

H = 512


def _run(scenes, mode, priors=None, pcl=None, seed=11, H=H, samples=None, thresh=0.5, minimal=8):
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
                               pcl=cu(pcl), H=H, seed=seed, samples=cu(samples), debug=True, minimal=minimal)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}, offs


def _compare(scenes, mode, priors=None, pcl=None, seed=11, minimal=8, sane=True):
    from oracle import solver as osv
    got, offs = _run(scenes, mode, priors, pcl, seed, minimal=minimal)
    for b, (k0, k1, K, Rgt, tgt) in enumerate(scenes):
        pr = None if priors is None else priors[b]
        solver = {'ransac': 'ransac', 'prior': 'prior_ransac', 'noprior': 'prior_ransac_noprior'}[mode]
        ret, nafter, tight, ultra, dbg = osv.estimate_pose(k0, k1, K, K, 0.5, solver=solver, priorRT=pr, seed=seed,
                                                            pair=b, H=H, pcl=pcl, minimal=minimal)
        if not dbg:                                                     # fewer than 5 correspondences (metrics.py:83-85)
            assert got['status'][b] == 0 and got['num_after'][b] == 0
            continue
        ss = dbg['samples'].shape[1]
        if ss == got['samples'].shape[2]:                               # (a 5..7-match pair of an 8-point batch keeps no sample dump)
            np.testing.assert_array_equal(got['samples'][b][:len(dbg['samples'])], dbg['samples'])   # integer path: bit exact
        v = dbg['valid']
        nm = len(v)                                                     # models of this pair (five-point: 10 per sample)
        # well-conditioned hypotheses agree to float64 round-off; judge on the Sampson counts, which is what
        # the selection consumes, and on F for the winning model
        same = got['count_all'][b][:nm][v] == dbg['count'][v]
        assert same.mean() > 0.995, same.mean()
        assert np.all(np.isinf(got['score_all'][b][:nm][~v])) and np.all(np.isinf(got['score_all'][b][nm:]))
        sl = slice(offs[b], offs[b + 1])
        if 'best' not in dbg:                                           # no valid model
            assert got['status'][b] == 0
            continue
        if not (ret is None and got['status'][b] == 0 and got['best'][b] < 0):
            assert got['best'][b] == dbg['best']
            np.testing.assert_allclose(got['E'][b], dbg['F'][dbg['best']], rtol=1e-7, atol=1e-9)
        assert (ret is not None) == bool(got['status'][b])
        if ret is not None:
            R, t, mask, E = ret
            np.testing.assert_array_equal(got['mask'][sl].astype(bool), mask)      # inlier mask: bit exact
            assert np.linalg.norm(got['R'][b] - R) < 1e-8 and np.linalg.norm(got['t'][b] - t) < 1e-8
            # and it is a sane pose
            if sane:
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
    s_four = tuple(a[:4] if i < 2 else a for i, a in enumerate(two_view_scene(50, seed=6)))
    got, offs = _run([s_few, s_ok, s_none, s_four], 'ransac')
    # 6 correspondences with outliers among them: five-point hypotheses are formed (the reference's gate is 5), but no model
    # collects more than the score floor -- the oracle decides, the kernel must agree (_compare below); 4 and 0: never a fit
    assert list(got['status'])[1:] == [1, 0, 0]
    assert got['num_after'][2] == 0 and got['num_after'][3] == 0
    _compare([s_few, s_ok, s_none, s_four], 'ransac')
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


# ---------------------------------------------------------------------------------------------------------------------
# five-point minimal solver (far_amd/csrc/solver5_f64.inc vs oracle/fivepoint.py)
# ---------------------------------------------------------------------------------------------------------------------
def test_five_point_models_equal_the_oracle_operation_for_operation():
    """Explicit 5-samples to both sides: the kernel restates oracle/fivepoint.py operation for operation in float64 without
    fp contraction, so the models (up to ten per sample), their validity and therefore every count agree -- reported as the
    fraction of bit-identical models, held to 1e-9 relative on all of them."""
    from oracle import fivepoint as fp
    from oracle import solver as osv
    from tests.util import planar_scene
    Hm = 640
    for kind, seed in (('general', 1), ('two_planes', 2), ('plane', 3)):
        p0, p1, R, t = planar_scene(400, seed, kind)
        Kc = np.array([[517.97, 0, 320.], [0, 517.97, 240.], [0, 0, 1.]])
        sc = (p0, p1, Kc, R, t)
        rng = np.random.default_rng(seed)
        samples = np.stack([rng.choice(len(p0), 5, replace=False) for _ in range(Hm // 10)]).astype(np.int32)
        samples[3, 4] = samples[3, 0]                                   # a sample that repeats a correspondence: rejected
        got, _ = _run([sc], 'ransac', samples=samples[None], H=Hm, minimal=5)
        kn0, kn1 = osv.normalize_keypoints(p0, p1, Kc, Kc)
        kp1 = kn0.astype(np.float32).astype(np.float64)
        kp2 = kn1.astype(np.float32).astype(np.float64)
        E, valid = fp.five_point(kp1[samples], kp2[samples])
        valid[3] = False
        diag = np.abs(np.stack([E[..., 0, 0], E[..., 1, 1], E[..., 2, 2]], -1)).min(-1)
        valid &= diag > 1e-4
        E = np.where(valid[..., None, None], E, 0.0).reshape(Hm, 3, 3)
        gv = np.isfinite(got['score_all'][0])
        np.testing.assert_array_equal(gv, valid.reshape(-1))
        F = got['F_all'][0]
        ident = (F == E).all((1, 2))
        print(f'[5pt kernel] {kind}: {int(valid.sum())} models of {Hm // 10} samples, bit-identical to the oracle: {100 * ident.mean():.2f} %')
        np.testing.assert_allclose(F, E, rtol=1e-9, atol=1e-12)
        assert ident.mean() > 0.99


def _g17():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g17_fivepoint.npz'))


def test_five_point_kernel_models_match_the_reference_solver():
    """k_hypotheses5 against the reference's own run_5point_our_kornia (cv_geometry.py:861-1043, golden G17a) on the committed
    five-point samples: the samples of a kind are packed as one 'pair' of 5 n correspondences with explicit sample indices
    (identity intrinsics: the points are already calibrated), the bars are those of the oracle test."""
    from tests.test_oracle_golden import G17_BARS, g17_model_agreement
    g = _g17()
    Ki = np.eye(3)
    for kind, (min_well, min_frac) in G17_BARS.items():
        p1, p2 = g[f'{kind}_p1'], g[f'{kind}_p2']
        n = len(p1)
        sc = (p1.reshape(-1, 2).astype(np.float32), p2.reshape(-1, 2).astype(np.float32), Ki, np.eye(3), np.array([1.0, 0, 0]))
        samples = np.arange(5 * n, dtype=np.int32).reshape(n, 5)
        got, _ = _run([sc], 'ransac', samples=samples[None], H=10 * n, minimal=5)
        F = got['F_all'][0].reshape(n, 10, 3, 3)
        valid = (np.abs(F).max((-1, -2)) > 0)
        nrm = np.linalg.norm(F, axis=(-1, -2), keepdims=True)
        E = np.where(valid[..., None, None], F / np.where(nrm > 0, nrm, 1.0), 0.0)
        # float32 keypoints here (the kernel's contract) against the float64 samples of the golden: 6e-8 relative on the points
        # times the conditioning of a five-point sample (measured: up to 5e-6 on the models) -> bar 2e-5; the kernel == oracle bit
        # for bit on identical inputs is test_five_point_models_equal_the_oracle_operation_for_operation, the oracle == reference
        # at 1e-6 on the float64 samples is tests/test_oracle_golden.py
        m = g17_model_agreement(kind, E, valid, g, tol=2e-5, drop_small_diagonal=True)
        print(f'[g17 kernel] {kind}: {m}')
        assert m['well'] >= min_well - 2 and m['frac_1e-6'] >= min_frac - 0.05, m


@pytest.mark.parametrize('tag,mode', [('p', 'prior'), ('n', 'noprior')])
def test_five_point_kernel_vs_reference_loop(tag, mode):
    """far_solver_f64 with five-point hypotheses on the committed sample indices against the reference's own
    RANSAC(model_type='essential').forward (golden G17b)."""
    from tests.test_oracle_golden import g17_loop_expectations
    g = _g17()
    scene = (g[f'{tag}_kpts0'], g[f'{tag}_kpts1'], g[f'{tag}_K'], g[f'{tag}_R_gt'], g[f'{tag}_t_gt'])
    priors = g['p_prior'][None] if tag == 'p' else None
    pcl = g['p_pcl'] if tag == 'p' else None
    smp = g[f'{tag}_samples'].astype(np.int32)[None]
    got, offs = _run([scene], mode, priors, pcl, H=10 * smp.shape[1], samples=smp, minimal=5)
    valid = np.isfinite(got['score_all'][0])
    best = int(got['best'][0])
    dE, sym = g17_loop_expectations(tag, best, valid, got['count_all'][0], (int(got['num_after'][0]), int(got['tight'][0]), int(got['ultra'][0])),
                                    got['mask'].astype(bool), got['F_all'][0][best], g)
    print(f'[g17 kernel loop] {tag}: winning model {dE:.2e} from the reference\'s, {sym} mask entries differ')
    assert np.linalg.norm(got['R'][0] - g[f'{tag}_R_gt']) < 0.05   # sanity only (measured 0.034)


def test_five_point_ransac_matches_oracle():
    """The whole loop with five-point hypotheses (minimal = 5): samples, counts, best model, masks, pose vs the oracle, in the
    plain and the prior branch, on general and two-plane scenes (where the 8-point's worst case is tens of degrees off)."""
    from tests.util import planar_scene
    Kc = np.array([[517.97, 0, 320.], [0, 517.97, 240.], [0, 0, 1.]])
    scenes = []
    for s, (M, kind) in enumerate([(300, 'general'), (500, 'two_planes'), (150, 'two_planes'), (64, 'general')]):
        p0, p1, R, t = planar_scene(M, 40 + s, kind)
        scenes.append((p0, p1, Kc, R, t))
    _compare(scenes, 'ransac', minimal=5)
    rng = np.random.default_rng(2)
    pcl = rng.uniform(-3.0, 3.0, (300, 3)).astype(np.float32)
    priors = []
    for (_, _, _, R, t) in scenes:
        d = 0.05 * rng.standard_normal(3)
        Rn = R @ (np.eye(3) + np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]]))
        priors.append(np.concatenate([Rn, (2.5 * t + 0.05 * rng.standard_normal(3))[:, None]], 1))
    _compare(scenes, 'prior', np.stack(priors), pcl, minimal=5)


def test_pairs_with_five_to_seven_matches_get_five_point_hypotheses():
    """An 8-point batch (minimal = 8) in which some pairs have only 5, 6 or 7 correspondences: the reference accepts them
    (metrics.py:83-85) and its executed solver fits them; here they take the five-point branch inside the same launches."""
    from tests.util import planar_scene
    Kc = np.array([[517.97, 0, 320.], [0, 517.97, 240.], [0, 0, 1.]])
    scenes = []
    for s, M in enumerate([7, 300, 6, 5, 8]):
        p0, p1, R, t = planar_scene(M, 60 + s, 'general', noise=0.05, outl=0.0)
        assert len(p0) == M
        scenes.append((p0, p1, Kc, R, t))
    got, _ = _run(scenes, 'ransac')
    print('[5..7 matches] status', list(got['status']), 'inliers', list(got['num_after']))
    assert list(got['status']) == [1, 1, 1, 0, 0]       # 5: the floor of 5 is not exceeded; 8 exact: the 8-point's floor of 8 neither
    _compare(scenes, 'ransac', sane=False)
    from oracle import metrics as om
    for b in (0, 2):
        T = np.eye(4); T[:3, :3] = scenes[b][3]; T[:3, 3] = scenes[b][4]
        te, Re, _ = om.relative_pose_error(T, got['R'][b], got['t'][b])
        assert Re < 1.0 and te < 5.0, (b, Re, te)


def test_planar_scenes_eight_point_vs_five_point():
    """What the solver choice means on the scenes the 8-point is degenerate on: success rate and pose error of both minimal
    solvers on general / two-plane / single-plane synthetic pairs (reported; the five-point bars are asserted)."""
    from oracle import metrics as om
    from tests.util import planar_scene
    Kc = np.array([[517.97, 0, 320.], [0, 517.97, 240.], [0, 0, 1.]])
    for kind in ('general', 'two_planes', 'plane'):
        scenes = []
        for s in range(16):
            p0, p1, R, t = planar_scene(300, 100 + s, kind)
            scenes.append((p0, p1, Kc, R, t))
        for minimal in (8, 5):
            got, _ = _run(scenes, 'ransac', H=2048, minimal=minimal)
            Re, te = [], []
            for b, sc in enumerate(scenes):
                T = np.eye(4); T[:3, :3] = sc[3]; T[:3, 3] = sc[4]
                e = om.relative_pose_error(T, got['R'][b], got['t'][b])
                Re.append(e[1]); te.append(e[0])
            print(f'[planar] {kind:10s} minimal={minimal}: fits {int(got["status"].sum())}/16, rotation error median {np.median(Re):.2f} max {np.max(Re):.2f} deg, '
                  f'translation-direction error median {np.median(te):.2f} max {np.max(te):.2f} deg')
            if minimal == 5 and kind != 'plane':        # (a two-plane scene whose second plane holds a handful of points stays hard)
                assert got['status'].all() and np.median(Re) < 0.5 and np.median(te) < 2.5 and np.mean(np.array(te) < 6.0) > 0.85


def test_prior_from_pose_matches_the_torch_formula():
    """K11c: the head's normalised 9-vector -> [R | t] prior (loftr.py:186-192) in one launch, against far_amd.pose6d's torch ops
    (rotation_6d_to_matrix after de-normalisation) in float64; R orthonormal; a degenerate 6D vector does not produce NaN."""
    from far_amd import ops
    from far_amd.pose6d import pose_mean_6d, pose_std_6d, rotation_6d_to_matrix
    g = torch.Generator(device='cuda').manual_seed(4)
    p = torch.randn(37, 9, device='cuda', generator=g) * 2.0
    mean, std = pose_mean_6d.cuda(), pose_std_6d.cuda()
    got = ops.prior_from_pose(p, mean, std)
    pd = p.double().cpu()
    R = rotation_6d_to_matrix(pd[:, 3:] * pose_std_6d[3:].double() + pose_mean_6d[3:].double())
    t = pd[:, :3] * pose_std_6d[:3].double() + pose_mean_6d[:3].double()
    ref = torch.cat([R, t[:, :, None]], -1)
    assert float((got.double().cpu() - ref).abs().max()) < 2e-6
    RtR = got[:, :, :3] @ got[:, :, :3].transpose(1, 2)
    assert float((RtR - torch.eye(3, device='cuda')).abs().max()) < 1e-5
    z = torch.zeros(1, 9, device='cuda')
    z[0, 3:] = -(mean[3:] / std[3:])                      # de-normalises to the zero 6D vector
    assert torch.isfinite(ops.prior_from_pose(z, mean, std)).all()
