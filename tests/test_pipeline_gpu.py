"""End-to-end parity on the GPU: far_amd.LoFTR + the evaluation call order vs the reference goldens and vs the
oracle's full path, on synthetic 640x480 pairs with the synthetic checkpoint."""
import json
import os

import numpy as np
import pytest
import torch

from far_amd import synth
from far_amd.config import far_eval_config
from tests.util import deviation

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def model():
    from far_amd.loftr import LoFTR
    m = LoFTR(far_eval_config()).eval()
    synth.load_synthetic(m, seed=0)
    return m.cuda()


def _batch(N, seed):
    im0, im1 = synth.synth_image_pair(N, seed=seed)
    K = torch.from_numpy(np.stack([synth.MP3D_K] * N)).cuda()
    return {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(),
            'dataset_name': ['mp3d']}, im0, im1


def test_matcher_vs_reference_golden(model):
    """LoFTR.forward on one 640x480 pair against the reference's own run (G7).  Bars = ~3x the deviation measured on
    MI355X between the fp32-grade kernels and the fp32 reference (deviation() prints the measured values)."""
    g = np.load(os.path.join(G, 'g7_full.npz'))
    data, _, _ = _batch(1, 0)
    with torch.no_grad():
        model(data)
    deviation('g7 feats_c', data['feats_c'][:, ::16, ::7, ::9], g['feats_c_sample'], atol=6e-5, rtol=1e-4)
    deviation('g7 featmap_f0', data['featmap_f0'][:, ::16, ::31, ::37], g['featmap_f0_sample'], atol=1.2e-4, rtol=1e-4)
    deviation('g7 featmap0 (tokens)', data['featmap0'][0, ::97], g['featmap0_sample'], atol=7e-5, rtol=1e-4)
    deviation('g7 featmap1 (tokens)', data['featmap1'][0, ::97], g['featmap1_sample'], atol=7e-5, rtol=1e-4)
    gi, gj = data['i_ids'].cpu().numpy(), data['j_ids'].cpu().numpy()
    got = dict(zip(gi.tolist(), gj.tolist()))
    ref = dict(zip(g['i_ids'].tolist(), g['j_ids'].tolist()))
    safe = (np.abs(g['rowmax'] - 0.2) > 1e-4) & (g['rowgap'] > 1e-4)            # SURVEY 8c protocol: margin 1e-4
    print('[g7] rows with margin:', int(safe.sum()), 'of', safe.size)
    for i in np.nonzero(safe)[0]:
        assert (i in got) == (i in ref), i
        if i in got:
            assert got[i] == ref[i]
    common = [i for i in ref if i in got]
    assert len(common) > 0.99 * len(ref) > 1000
    a = np.array([{i: n for n, i in enumerate(gi.tolist())}[i] for i in common])
    b = np.array([{i: n for n, i in enumerate(g['i_ids'].tolist())}[i] for i in common])
    deviation('g7 mconf', data['mconf'][a], g['mconf'][b], atol=1.5e-4, rtol=0)
    deviation('g7 mkpts1_f', data['mkpts1_f'][a], g['mkpts1_f'][b], atol=4e-3, rtol=0)
    deviation('g7 expec_f', data['expec_f'][a], g['expec_f'][b], atol=1.5e-3, rtol=0)


def test_head_vs_reference_golden(model):
    g = np.load(os.path.join(G, 'g4_head.npz'))
    rng = np.random.default_rng(14)
    f0 = rng.standard_normal((1, 4800, 256)).astype(np.float32)
    f1 = (0.5 * f0 + rng.standard_normal((1, 4800, 256))).astype(np.float32)
    n = g['counts']
    data = {'featmap0': torch.from_numpy(f0).cuda(), 'featmap1': torch.from_numpy(f1).cuda(),
            'loftr_rt': torch.from_numpy(g['loftr_rt']).cuda(), 'num_correspondences': torch.tensor([int(n[0])]).cuda(),
            'num_correspondences_before_ransac': torch.tensor([int(n[1])]).cuda(),
            'inliers_best_tight': torch.tensor([int(n[2])]).cuda(), 'inliers_best_ultra_tight': torch.tensor([int(n[3])]).cuda()}
    with torch.no_grad():
        emm = model.loftr_regress.emm
        blk = emm(torch.cat([data['featmap0'], data['featmap1']], 0)).cpu().numpy()
        _, _, _, _, lp, ilp = model.preprocess_helper(data)
        model.forward_rt_prediction(data)
    sc = np.abs(g['block_out']).max()
    np.testing.assert_allclose(blk, g['block_out'], atol=2e-4 * sc, rtol=1e-3)
    np.testing.assert_allclose(lp.cpu().numpy(), g['loftr_preds_6d'], atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(ilp.cpu().numpy(), g['inv_loftr_preds_6d'], atol=1e-5, rtol=1e-5)
    reg = data['regressed_rt'].cpu().numpy()
    assert reg.shape == (1, 9) and data['expec_rt'].shape == (9,) and data['priorRT'].shape == (3, 4)
    # north_star: regression logits within 1e-3 relative (fp32)
    np.testing.assert_allclose(reg, g['regressed_rt'], atol=1e-3 * np.abs(g['regressed_rt']).max(), rtol=1e-3)
    np.testing.assert_allclose(data['priorRT'], g['priorRT'], atol=2e-3, rtol=1e-3)


def test_full_step_vs_oracle(model):
    """matcher -> solver -> head -> solver(prior) -> head on 2 pairs.  Each stage is checked against the oracle
    on the SAME inputs (the GPU's own upstream outputs): discrete decisions (RANSAC argmax) are only
    comparable that way; the matcher stage is compared set-wise against the oracle's own matcher."""
    from far_amd.config import RunCfg
    from far_amd.supervision import compute_supervision_RT
    from oracle import head as oh
    from oracle import model as om
    from oracle import solver as osv
    cfg = far_eval_config()
    data, im0, im1 = _batch(2, 21)
    run = RunCfg('prior_ransac', 2)
    Hn, seed = 512, 3
    man = json.load(open(os.path.join(G, 'g8_state_dict_manifest.json')))
    w = om.Weights(synth.synthetic_state_dict({k: tuple(v) for k, v in man.items()}))
    pcl = np.random.RandomState(0).uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
    pos = oh.positional_encodings()
    K = synth.MP3D_K
    with torch.no_grad():
        model(data)
    odata = om.matcher_forward(w, cfg, im0, im1)
    gm = set(zip(data['b_ids'].tolist(), data['i_ids'].tolist(), data['j_ids'].tolist()))
    rm = set(zip(odata['b_ids'].tolist(), odata['i_ids'].tolist(), odata['j_ids'].tolist()))
    assert len(gm & rm) > 0.99 * len(rm) > 2000
    np.testing.assert_allclose(data['featmap0'].cpu().numpy(), odata['featmap0'], atol=5e-3, rtol=1e-3)
    mk0, mk1 = data['mkpts0_f'].cpu().numpy(), data['mkpts1_f'].cpu().numpy()
    bids = data['m_bids'].cpu().numpy()
    f0, f1 = data['featmap0'].cpu().numpy(), data['featmap1'].cpu().numpy()
    prior = None
    for rnd in range(2):
        data['translation_scale'] = None
        compute_supervision_RT(data, run, H=Hn, seed=seed)
        rt = data['loftr_rt'].cpu().numpy()
        assert rt.shape == (2, 3, 4) and data['loftr_rt'].dtype == torch.float64
        cnt = {k: data[k].cpu().numpy() for k in ['num_correspondences', 'num_correspondences_before_ransac',
                                                  'inliers_best_tight', 'inliers_best_ultra_tight']}
        mask = data['solver_inlier_mask'].cpu().numpy().astype(bool)
        for b in range(2):
            sel = bids == b
            ret, na, ti, ul, _ = osv.estimate_pose(mk0[sel], mk1[sel], K, K, 0.5, solver='prior_ransac',
                                                   priorRT=None if prior is None else prior[b], seed=seed, pair=b,
                                                   H=Hn, pcl=pcl)
            R, t, m, E = ret
            assert np.linalg.norm(rt[b] - np.concatenate([R, t[:, None]], 1)) < 1e-4      # north_star: 1e-4 Frobenius
            np.testing.assert_array_equal(mask[sel], m)                                     # bit-exact inlier mask
            assert (cnt['num_correspondences'][b], cnt['inliers_best_tight'][b], cnt['inliers_best_ultra_tight'][b]) \
                == (na, ti, ul)
            assert cnt['num_correspondences_before_ransac'][b] == sel.sum()
            # (pose sanity against ground truth lives in test_solver_gpu.py on real 3-D scenes; the banded
            #  image pairs here have the usual small-baseline translation/rotation ambiguity)
            assert abs(np.linalg.det(R) - 1) < 1e-9 and abs(np.linalg.norm(t) - 1) < 1e-9
        with torch.no_grad():
            model.forward_rt_prediction(data)
        reg = data['regressed_rt'].cpu().numpy()
        prior = np.asarray(data['priorRT'])
        assert reg.shape == (2, 9) and prior.shape == (2, 3, 4)
        for b in range(2):
            lp, ilp = om.preprocess_helper(cfg, rt[b], cnt['num_correspondences'][b],
                                           cnt['num_correspondences_before_ransac'][b], cnt['inliers_best_tight'][b],
                                           cnt['inliers_best_ultra_tight'][b])
            oreg, gate, _ = om.head_forward(w, cfg, f0[b:b + 1], f1[b:b + 1], lp, ilp, pos)
            np.testing.assert_allclose(reg[b], oreg[0], atol=1e-3 * np.abs(oreg).max(), rtol=1e-3)
            np.testing.assert_allclose(prior[b], om.prior_from_regressed(oreg), atol=2e-3, rtol=1e-3)
    # data-dict contract
    for k in ['conf_matrix', 'b_ids', 'i_ids', 'j_ids', 'gt_mask', 'm_bids', 'mkpts0_c', 'mkpts1_c', 'mconf', 'W',
              'expec_f', 'mkpts0_f', 'mkpts1_f', 'featmap0', 'featmap1', 'mask_c0', 'mask_c1', 'translation_scale',
              'loftr_rt', 'expec_rt', 'expec_e', 'num_correspondences', 'num_correspondences_before_ransac',
              'num_correspondences_after_ransac', 'inliers_best_tight', 'inliers_best_ultra_tight', 'regressed_rt',
              'priorRT', 'bs', 'hw0_i', 'hw0_c', 'hw0_f']:
        assert k in data, k


def test_pipeline_call_order_runs(model):
    from far_amd.pipeline import test_step
    data, _, _ = _batch(3, 5)
    test_step(model, data, H=256)
    assert data['regressed_rt'].shape == (3, 9) and torch.isfinite(data['regressed_rt']).all()
    assert data['loftr_rt'].shape == (3, 3, 4)


def test_head_prefetch_changes_nothing_and_is_only_for_callers_that_run_the_head(model):
    """pipeline.test_step lets the head's feature stage run behind the coarse matcher (LoFTR.head_prefetch): the same results to
    the bit as without it; a matcher-only caller -- here at a resolution the head's 60 x 80 position table does not fit, the
    Map-free use -- gets no head work at all."""
    from far_amd.pipeline import test_step
    d1, _, _ = _batch(2, 7)
    test_step(model, d1, H=256)
    assert '_far_head_follows' not in d1
    model.head_prefetch = False
    try:
        d2, _, _ = _batch(2, 7)
        test_step(model, d2, H=256)
    finally:
        del model.head_prefetch                                  # back to the class default
    for k in ('i_ids', 'j_ids', 'mconf', 'mkpts1_f', 'regressed_rt', 'loftr_rt', 'priorRT'):
        a, b = d1[k], d2[k]
        assert (torch.equal(a, b) if torch.is_tensor(a) else np.array_equal(a, b)), k
    im0, im1 = synth.synth_image_pair(1, seed=3)
    small = {'image0': torch.from_numpy(np.ascontiguousarray(im0[:, :, :240, :320])).cuda(),
             'image1': torch.from_numpy(np.ascontiguousarray(im1[:, :, :240, :320])).cuda()}
    with torch.no_grad():
        model(small)                                             # 30 x 40 coarse tokens: the head could not take them
    assert model._HEAD_KEY not in small and small['mkpts0_f'].shape[1] == 2


def test_cached_prediction_mode_head_only():
    """BASELINE config 4 / `--from_saved_preds`: the matcher is not constructed (loftr.py:20); the dataset supplies
    featmap0/1 + the solver outputs, and only forward_rt_prediction runs (lightning_loftr.py:326,334)."""
    from far_amd.loftr import LoFTR
    cfg = far_eval_config()
    cfg['from_saved_preds'] = 'loftr_preds'
    m = LoFTR(cfg).eval()
    assert not hasattr(m, 'backbone') and not hasattr(m, 'coarse_matching')
    man = json.load(open(os.path.join(G, 'g8_state_dict_manifest.json')))
    sd = synth.synthetic_state_dict({k: tuple(v) for k, v in man.items() if k.startswith('loftr_regress.')})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda()
    g = np.load(os.path.join(G, 'g4_head.npz'))
    rng = np.random.default_rng(14)
    f0 = rng.standard_normal((1, 4800, 256)).astype(np.float32)
    f1 = (0.5 * f0 + rng.standard_normal((1, 4800, 256))).astype(np.float32)
    B = 3
    n = g['counts']
    rep = lambda a: torch.from_numpy(a).cuda().repeat(B, 1, 1)
    data = {'featmap0': rep(f0), 'featmap1': rep(f1), 'loftr_rt': torch.from_numpy(g['loftr_rt']).cuda().repeat(B, 1, 1),
            'num_correspondences': torch.tensor([int(n[0])] * B).cuda(),
            'num_correspondences_before_ransac': torch.tensor([int(n[1])] * B).cuda(),
            'inliers_best_tight': torch.tensor([int(n[2])] * B).cuda(),
            'inliers_best_ultra_tight': torch.tensor([int(n[3])] * B).cuda()}
    with torch.no_grad():
        m.forward_rt_prediction(data)
    reg = data['regressed_rt'].cpu().numpy()
    assert reg.shape == (B, 9)
    for b in range(B):       # a batch of B is B independent B = 1 runs
        np.testing.assert_allclose(reg[b:b + 1], g['regressed_rt'], atol=1e-3 * np.abs(g['regressed_rt']).max(), rtol=1e-3)


def _head_inputs(rng, B=2):
    f0 = torch.from_numpy(rng.standard_normal((B, 4800, 256)).astype(np.float32)).cuda()
    f1 = torch.from_numpy(rng.standard_normal((B, 4800, 256)).astype(np.float32)).cuda()
    rt = torch.eye(3, 4, dtype=torch.float64).repeat(B, 1, 1).cuda()
    rt[:, :, 3] = torch.tensor([0.6, 0.0, 0.8], dtype=torch.float64)
    cnt = lambda v: torch.full((B,), v, dtype=torch.int64).cuda()
    return {'featmap0': f0, 'featmap1': f1, 'loftr_rt': rt, 'num_correspondences': cnt(700),
            'num_correspondences_before_ransac': cnt(1500), 'inliers_best_tight': cnt(400), 'inliers_best_ultra_tight': cnt(50)}


def test_head_feature_reuse_is_exact_and_never_stale(model):
    """The second head call of a step reuses the pair features (kept in the caller's data dict).  Reuse must be
    bit-identical to recomputing, and must NOT survive: a torch in-place edit, a raw-pointer overwrite through
    far_amd.ops(out=...) (round-1 hazard: kernels write behind torch's version counter), the same buffers handed
    over in a new dict, a weight change, a precision change.  It must also work under torch.inference_mode()."""
    import copy
    from far_amd import ops
    m = copy.deepcopy(model)
    head = m.loftr_regress
    rng = np.random.default_rng(3)
    data = _head_inputs(rng)
    key = m._HEAD_KEY
    with torch.no_grad():
        head.cache_features = False
        m.forward_rt_prediction(data)
        a1 = data['regressed_rt'].clone()
        assert key not in data
        data['loftr_rt'][:, 0, 3] = 0.3
        m.forward_rt_prediction(data)
        a2 = data['regressed_rt'].clone()
        head.cache_features = True
        data['loftr_rt'][:, 0, 3] = 0.6
        m.forward_rt_prediction(data)
        b1 = data['regressed_rt'].clone()
        feats = data[key][1]
        data['loftr_rt'][:, 0, 3] = 0.3
        m.forward_rt_prediction(data)                       # served from the dict
        b2 = data['regressed_rt'].clone()
        assert data[key][1] is feats
        assert torch.equal(a1, b1) and torch.equal(a2, b2) and not torch.equal(a1, a2)
        # (1) torch in-place edit
        data['featmap0'].add_(1.0)
        m.forward_rt_prediction(data)
        assert data[key][1] is not feats and not torch.equal(data['regressed_rt'], b2)
        # (2) overwrite through a kernel's out= (raw device pointer): the next call must see the new contents
        feats = data[key][1]
        fresh = torch.from_numpy(rng.standard_normal((2, 4800, 256)).astype(np.float32)).cuda()
        ops.layernorm(fresh, head.norm.weight, head.norm.bias, 1e-6, out=data['featmap0'])
        m.forward_rt_prediction(data)
        got = data['regressed_rt'].clone()
        assert data[key][1] is not feats
        ref = dict(data)
        ref.pop(key)
        ref['featmap0'] = data['featmap0'].clone()
        head.cache_features = False
        m.forward_rt_prediction(ref)
        head.cache_features = True
        assert torch.equal(got, ref['regressed_rt'])
        # (3) the same (preallocated, overwritten) buffers in a NEW dict: nothing carries over
        nxt = {k: v for k, v in data.items() if k != key}
        m.forward_rt_prediction(nxt)
        assert nxt[key][1] is not data[key][1] and torch.equal(nxt['regressed_rt'], got)
        # (4) weights change between two calls on one dict (scoring several checkpoints on cached predictions)
        feats = data[key][1]
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd['loftr_regress.emm.norm1.weight'] = sd['loftr_regress.emm.norm1.weight'] * 1.5
        m.load_state_dict(sd)
        m.forward_rt_prediction(data)
        assert data[key][1] is not feats and not torch.equal(data['regressed_rt'], got)
        # (5) operand precision change
        feats = data[key][1]
        head_only_layers = [x for x in head.modules() if hasattr(x, 'split_operands')]
        for x in head_only_layers:
            x.split_operands = False
        m.forward_rt_prediction(data)
        assert data[key][1] is not feats
        for x in head_only_layers:
            x.split_operands = True
    # (6) Lightning >= 1.8 runs test/validate/predict under inference_mode: tensors have no version counter
    with torch.inference_mode():
        d2 = _head_inputs(np.random.default_rng(3))
        m2 = copy.deepcopy(model)
        m2.forward_rt_prediction(d2)
        c1 = d2['regressed_rt'].clone()
        d2['loftr_rt'] = d2['loftr_rt'].clone()
        d2['loftr_rt'][:, 0, 3] = 0.3
        m2.forward_rt_prediction(d2)
        c2 = d2['regressed_rt'].clone()
    assert torch.equal(c1, a1) and torch.equal(c2, a2)


def test_full_step_under_inference_mode(model):
    """pipeline.test_step inside torch.inference_mode() (what Lightning's trainer.test does) == under no_grad."""
    from far_amd.pipeline import test_step
    d1, _, _ = _batch(2, 5)
    test_step(model, d1, H=256)
    with torch.inference_mode():
        d2, _, _ = _batch(2, 5)
        test_step(model, d2, H=256)
    for k in ('i_ids', 'j_ids', 'mconf', 'mkpts1_f', 'regressed_rt', 'loftr_rt'):
        assert torch.equal(d1[k], d2[k]), k


def test_batch32_is_32_independent_single_pair_runs(model):
    """BASELINE configs[1] runs 32 pairs per step; the reference's solver + head are batch-size-1 code (SURVEY.md
    section 0 fact 4), so "batch 32" MEANS 32 independent B = 1 runs stacked.  Every kernel of this library is
    row / pixel / problem independent, so pair b of the batch must equal the B = 1 run of pair b BIT FOR BIT through
    matcher, solver AND head (ids, confidences, sub-pixel positions, [R | t], inlier mask, counts, regressed_rt, priorRT):
    since round 3 the head's Linear layers, its LayerNorms and the 70 x N x 70 contraction run on K9 / K6 / K15, whose
    outputs depend on their own row only (round 2: vendor GEMMs chosen by row count, a 1e-4 bar)."""
    from far_amd.config import RunCfg
    from far_amd.supervision import compute_supervision_RT
    B, Hn, seed = 32, 512, 2
    data, _, _ = _batch(B, 77)
    run = RunCfg('prior_ransac', 2)
    with torch.no_grad():
        model(data)
        compute_supervision_RT(data, run, H=Hn, seed=seed)
        model.forward_rt_prediction(data)
    counts = [int(c) for c in data['match_counts']]
    offs = np.concatenate([[0], np.cumsum(counts)])
    assert min(counts) > 1000
    reg = data['regressed_rt']
    scale = float(reg.abs().max())
    worst = 0.0
    for b in range(B):
        d1 = {'image0': data['image0'][b:b + 1], 'image1': data['image1'][b:b + 1], 'K0': data['K0'][b:b + 1],
              'K1': data['K1'][b:b + 1], 'dataset_name': ['mp3d']}
        with torch.no_grad():
            model(d1)
            # the sampling hash is keyed by (seed XOR pair index, hypothesis, slot) (oracle/solver.py:hash_u32):
            # pair 0 of a run seeded with seed ^ b draws the samples of pair b of the batch
            compute_supervision_RT(d1, run, H=Hn, seed=seed ^ b)
            model.forward_rt_prediction(d1)
        sl = slice(int(offs[b]), int(offs[b + 1]))
        assert int(d1['match_counts'][0]) == counts[b], b
        for k in ('i_ids', 'j_ids', 'mconf', 'mkpts0_f', 'mkpts1_f', 'expec_f', 'solver_inlier_mask'):
            assert torch.equal(data[k][sl], d1[k]), (b, k)
        assert torch.equal(data['loftr_rt'][b], d1['loftr_rt']), b
        for k in ('num_correspondences', 'num_correspondences_before_ransac', 'inliers_best_tight', 'inliers_best_ultra_tight'):
            assert int(data[k][b]) == int(d1[k][0]), (b, k)
        assert torch.equal(data['featmap0'][b], d1['featmap0'][0]) and torch.equal(data['featmap1'][b], d1['featmap1'][0])
        worst = max(worst, float((reg[b] - d1['regressed_rt'][0]).abs().max()))
        assert torch.equal(reg[b], d1['regressed_rt'][0]), (b, reg[b], d1['regressed_rt'][0])
        assert np.array_equal(np.asarray(data['priorRT'])[b], np.asarray(d1['priorRT'])), b
    print(f'[batch32] max |regressed_rt(batch) - regressed_rt(single)| = {worst:.3e} (scale {scale:.3e})')
    assert worst == 0.0


def test_training_step_on_gpu(model, monkeypatch):
    """Training-mode forward + backward on the GPU against golden G10 from the reference.  K1 (sparse-position
    confidences), K5, the K9 Linear layers and K2 run their HIP forward AND backward kernels; the backbone and the
    fine-window ops are the differentiable vendor-op forms.  The two sampling draws of the reference
    (coarse_matching.py:216-229) are taken from the CPU generator here so that they equal the golden's; then the
    sampled ids, the three losses, regressed_rt and the 16 parameter-gradient norms of G10 must all be reproduced."""
    import copy
    from tests.test_training_cpu import _train_helpers
    h = _train_helpers()
    g = np.load(os.path.join(G, 'g10_training.npz'))
    m = copy.deepcopy(model)
    im0, im1, ii, jj, rt = h.train_inputs()
    data = {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(),
            'spv_b_ids': torch.zeros(len(ii), dtype=torch.int64).cuda(), 'spv_i_ids': torch.from_numpy(ii).cuda(),
            'spv_j_ids': torch.from_numpy(jj).cuda()}
    real_randint = torch.randint

    def randint_from_cpu_stream(*args, device=None, **kw):
        out = real_randint(*args, **kw)
        return out if device is None else out.to(device)
    monkeypatch.setattr(torch, 'randint', randint_from_cpu_stream)
    m.train()
    torch.manual_seed(123)
    m(data, train=True)
    monkeypatch.setattr(torch, 'randint', real_randint)
    # the GPU training path builds no dense conf_matrix: the confidences at the ground-truth positions come from K1's
    # training kernels (HIP forward + backward), far_amd/losses.py reads them
    assert data['conf_matrix'] is None and data['conf_pos'].requires_grad and data['expec_f'].requires_grad
    for k in ['b_ids', 'i_ids', 'j_ids']:
        np.testing.assert_array_equal(data[k].cpu().numpy(), g[k])
    assert len(data['mconf']) == int(g['n_mconf'])
    data.update({'loftr_rt': torch.from_numpy(rt).cuda(), 'num_correspondences': torch.tensor([731]).cuda(),
                 'num_correspondences_before_ransac': torch.tensor([1500]).cuda(),
                 'inliers_best_tight': torch.tensor([410]).cuda(), 'inliers_best_ultra_tight': torch.tensor([57]).cuda()})
    m.forward_rt_prediction(data)
    from far_amd.losses import coarse_positive_conf
    loss_c = -torch.log(coarse_positive_conf(data) + 1e-6).mean()
    loss_f = data['expec_f'].pow(2).mean()
    loss_rt = data['regressed_rt'].pow(2).sum()
    m.zero_grad()
    (loss_c + loss_f + loss_rt).backward()
    deviation('train losses', np.array([loss_c.item(), loss_f.item(), loss_rt.item()]), g['losses'], rtol=1e-3, atol=0)
    deviation('train expec_f', data['expec_f'][:64], g['expec_f_head'], atol=6e-4, rtol=0)
    deviation('train regressed_rt', data['regressed_rt'], g['regressed_rt'], atol=2e-3 * np.abs(g['regressed_rt']).max(), rtol=2e-3)
    P = dict(m.named_parameters())
    worst = 0.0
    for n, k in enumerate(h.GRAD_KEYS):
        gr = P[k].grad
        assert gr is not None, k
        rel = abs(gr.norm().item() - g['grad_norms'][n]) / g['grad_norms'][n]
        worst = max(worst, rel)
        print(f'[train grad] {k}: |g| {gr.norm().item():.6e} vs reference {g["grad_norms"][n]:.6e} (rel {rel:.2e})')
        s_ = gr.reshape(-1)[:: max(1, gr.numel() // 8)][:8].cpu().numpy()
        if os.environ.get('FAR_MEASURE_ONLY') != '1':
            np.testing.assert_allclose(gr.norm().item(), g['grad_norms'][n], rtol=5e-3, err_msg=k)
            np.testing.assert_allclose(s_, g['grad_samples'][n], rtol=2e-2, atol=4e-3 * np.abs(g['grad_samples'][n]).max() + 1e-7, err_msg=k)
    print(f'[train grad] worst relative deviation of a gradient norm: {worst:.2e}')
    n_grad = 0
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), k
            n_grad += 1
    assert n_grad == sum(1 for _ in m.parameters()) == 189      # every parameter receives a gradient


def test_train_step_pipeline_two_pairs_and_optimizer(model):
    """pipeline.train_step (the reference's _trainval_inference order) + LoFTRLoss + AdamW on a batch of two pairs: the
    HIP training path end to end.  Checks the data-dict contract of the training forward (sparse conf_pos, sampled ids,
    expec_f_gt, loss scalars), that every parameter receives a finite gradient, and that a few optimizer steps on the
    same batch reduce the loss (the gradients point downhill)."""
    import copy
    from far_amd.config import far_train_config, RunCfg
    from far_amd.losses import LoFTRLoss
    from far_amd.pipeline import train_step
    cfg = far_train_config()
    m = copy.deepcopy(model).train()
    loss_fn = LoFTRLoss(cfg).train()
    base = synth.synth_training_batch(2, seed=77, device='cuda')
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.1)
    losses = []
    for it in range(4):
        batch = dict(base)
        torch.manual_seed(5)
        train_step(m, batch, loss_fn, RunCfg('prior_ransac', 2), H=512, seed=0)
        if it == 0:
            assert batch['conf_matrix'] is None and batch['conf_pos'].shape == base['spv_b_ids'].shape
            assert batch['expec_f_gt'].shape == (len(batch['b_ids']), 2) and batch['expec_f'].shape[1] == 3
            assert batch['regressed_rt'].requires_grad and batch['loss'].requires_grad
            assert set(batch['loss_scalars']) >= {'loss', 'loss_c', 'loss_f', 'loss_rot', 'loss_tr'}
            n_train = int(2 * 4800 * m.coarse_matching.train_coarse_percent)
            assert len(batch['b_ids']) == n_train                      # predictions sampled / padded with ground truth (:205-240)
        batch['loss'].backward()
        if it == 0:
            for k, p_ in m.named_parameters():
                assert p_.grad is not None and torch.isfinite(p_.grad).all(), k
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(float(batch['loss']))
    print('[train_step] losses over 4 AdamW steps:', ' '.join(f'{x:.5f}' for x in losses))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_train_step_on_a_pair_without_predictions_or_ground_truth(model):
    """Early-training / bad-pair corner cases in one batch of two pairs, both blank images (the matcher predicts nothing):
    pair-level ground truth present (every sampled match is a padded GT match, coarse_matching.py:225-240; the solver gets
    M = 0 and falls back to the identity, supervision.py:221-224), and -- second run -- no ground truth at all (the dummy
    (0, 0, 0) entry with zero coarse weight, loftr_loss.py:65-70).  Loss and all gradients finite."""
    import copy
    from far_amd.config import far_train_config, RunCfg
    from far_amd.losses import LoFTRLoss
    from far_amd.pipeline import train_step
    m = copy.deepcopy(model).train()
    m.coarse_matching.thr = 2.0                   # confidences are <= 1: the matcher predicts nothing (training-mode BatchNorm
                                                  # still finds a few matches between two blank images otherwise)
    loss_fn = LoFTRLoss(far_train_config()).train()
    base = synth.synth_training_batch(2, seed=78, device='cuda')
    base['image0'] = torch.full_like(base['image0'], 0.5)
    base['image1'] = torch.full_like(base['image1'], 0.5)
    zero = torch.zeros(1, dtype=torch.int64, device='cuda')
    for no_gt in (False, True):
        batch = dict(base)
        if no_gt:
            batch.update(spv_b_ids=zero, spv_i_ids=zero, spv_j_ids=zero, spv_gt_count=0)
        m.zero_grad()
        train_step(m, batch, loss_fn, RunCfg('prior_ransac', 2), H=256, seed=0)
        assert int((batch['mconf'] != 0).sum()) == 0                       # nothing predicted: only padded ground truth
        assert batch['solver_status'].tolist() == [0, 0]
        np.testing.assert_array_equal(batch['loftr_rt'].cpu().numpy(), np.stack([np.eye(3, 4)] * 2))
        assert torch.isfinite(batch['loss']).all()
        if no_gt:
            assert float(batch['loss_scalars']['loss_c']) == 0.0
        batch['loss'].backward()
        for k, p_ in m.named_parameters():
            assert p_.grad is None or torch.isfinite(p_.grad).all(), k


def test_step_with_no_match_in_the_whole_batch(model):
    """Every pair blank (B = 2 and B = 1): M = 0 everywhere -- empty (0, .) tensors from the fine stage
    (fine_preprocess.py:34-37, fine_matching.py:33-41), a solver batch without a single correspondence, identity poses into
    the head; nothing raises, the head's output is finite."""
    from far_amd.pipeline import test_step
    for B in (2, 1):
        data, _, _ = _batch(B, 9)
        data['image0'].zero_()
        data['image1'].zero_()
        test_step(model, data, H=256)
        assert data['b_ids'].numel() == 0 and data['mkpts0_f'].shape == (0, 2) and data['mkpts1_f'].shape == (0, 2)
        assert data['expec_f'].shape == (0, 3)
        assert data['solver_status'].tolist() == [0] * B
        rt = data['loftr_rt'].reshape(B, 3, 4).cpu().numpy()
        np.testing.assert_array_equal(rt, np.stack([np.eye(3, 4)] * B))
        assert data['regressed_rt'].shape == (B, 9) and torch.isfinite(data['regressed_rt']).all()


def test_step_on_a_side_stream_equals_the_default_stream(model):
    """Every far_* call is enqueued on torch's CURRENT stream (the stream argument of the C ABI); nothing may silently run
    on the null stream.  The whole step on a side stream must reproduce the default-stream results exactly."""
    from far_amd.pipeline import test_step
    ref, _, _ = _batch(2, 12)
    test_step(model, ref, H=256)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    data, _, _ = _batch(2, 12)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        test_step(model, data, H=256)
    side.synchronize()
    for k in ('b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts1_f', 'loftr_rt', 'regressed_rt', 'solver_inlier_mask'):
        assert torch.equal(data[k], ref[k]), k


def test_side_streams_change_nothing(model):
    """The head's feature stage on a second stream (LoFTR.head_side_stream, default on) and the FPN's fine branch on a third
    (LoFTR.fpn_side_stream, opt-in): independent branches of the step's dependency graph -- the same results to the bit in every
    combination, on the default stream and with the whole step on a caller's stream, twice in a row (the second step reuses the
    blocks the first one's side streams freed)."""
    from far_amd.pipeline import test_step
    keys = ('b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts0_f', 'mkpts1_f', 'loftr_rt', 'regressed_rt', 'solver_inlier_mask', 'priorRT')

    def run(head, fpn, stream=None):
        model.head_side_stream, model.fpn_side_stream = head, fpn
        try:
            outs = []
            for _ in range(2):
                d, _, _ = _batch(8, 21)
                torch.cuda.synchronize()
                with torch.cuda.stream(stream or torch.cuda.current_stream()):
                    test_step(model, d, H=256)
                torch.cuda.synchronize()
                assert '_side_pending' not in model.__dict__ and '_fpn_pending' not in model.__dict__      # everything joined
                outs.append({k: (d[k].clone() if torch.is_tensor(d[k]) else np.array(d[k])) for k in keys})
            return outs
        finally:
            del model.head_side_stream, model.fpn_side_stream

    ref = run(False, False)[0]
    for head, fpn, stream in ((True, False, None), (False, True, None), (True, True, None), (True, True, torch.cuda.Stream())):
        for got in run(head, fpn, stream):
            for k in keys:
                a, b = ref[k], got[k]
                assert (torch.equal(a, b) if torch.is_tensor(a) else np.array_equal(a, b)), (head, fpn, stream is not None, k)


def test_a_step_leaves_no_device_memory_to_the_cyclic_collector(model):
    """Round 6: two closures that referred to themselves (`run.again = ...`) made every step's data dict part of a reference cycle: 2.9 GiB
    of device tensors per 32-pair step stayed allocated until Python's generation-2 collector came by (3.5 -> 30 GiB over ten steps with the
    collector off), the caching allocator grew by hipMalloc in the middle of timed steps and bench lines jumped by 30 ms now and then.
    With the collector DISABLED the allocated device memory must be flat from step to step."""
    import gc
    from far_amd.pipeline import test_step

    def step():
        d, _, _ = _batch(4, 31)
        test_step(model, d, H=256)
        torch.cuda.synchronize()

    for _ in range(2):
        step()
    gc.collect()
    gc.disable()
    try:
        mem = []
        for _ in range(6):
            step()
            mem.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert max(mem) - min(mem) < (8 << 20), [round(x / 2**20, 1) for x in mem]


def test_non_contiguous_inputs_are_accepted(model):
    """Images that are strided views (a pair tensor (N, 2, H, W) sliced per view, as a dataloader may hand them over) and
    intrinsics that are views of a larger tensor: same results as with packed copies."""
    from far_amd.pipeline import test_step
    ref, im0, im1 = _batch(2, 13)
    test_step(model, ref, H=256)
    pair = torch.from_numpy(np.concatenate([im0, im1], 1)).cuda()              # (N, 2, H, W)
    Kbig = torch.zeros(2, 2, 3, 3, dtype=torch.float64, device='cuda')
    Kbig[:, 0] = ref['K0']
    Kbig[:, 1] = ref['K1']
    data = {'image0': pair[:, 0:1], 'image1': pair[:, 1:2], 'K0': Kbig[:, 0], 'K1': Kbig[:, 1], 'dataset_name': ['mp3d']}
    assert not data['image0'].is_contiguous() and not data['K0'].is_contiguous()
    test_step(model, data, H=256)
    for k in ('b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts1_f', 'loftr_rt', 'regressed_rt'):
        assert torch.equal(data[k], ref[k]), k


def test_pair_without_matches_in_a_batch(model):
    """A blank pair inside a batch: no coarse matches for it, the solver reports failure for that pair and the
    reference's identity fallback applies (supervision.py:221-224); the other pair is unaffected."""
    from far_amd.pipeline import test_step
    data, _, _ = _batch(2, 9)
    solo, _, _ = _batch(2, 9)
    data['image0'][1].zero_()
    data['image1'][1].zero_()
    test_step(model, data, H=256)
    counts = [int(c) for c in data['match_counts']]
    assert counts[0] > 1000 and counts[1] == 0
    rt = data['loftr_rt'].cpu().numpy()
    np.testing.assert_array_equal(rt[1], np.concatenate([np.eye(3), np.zeros((3, 1))], 1))
    assert int(data['solver_status'][1]) == 0 and int(data['num_correspondences'][1]) == 0
    assert torch.isfinite(data['regressed_rt']).all()
    # pair 0 of the mixed batch == pair 0 of a normal batch (pairs are independent problems): every kernel of this
    # library is row/pixel-independent, so the coarse stage is bit-identical; the fine transformer's vendor GEMMs pick
    # their kernel by the TOTAL window count, which moves sub-pixel positions by an ulp (and, through RANSAC's
    # discontinuity, the solver pose by more) -- compared with a tolerance.
    test_step(model, solo, H=256)
    n0 = counts[0]
    assert int(solo['match_counts'][0]) == n0
    for k in ('i_ids', 'j_ids', 'mconf', 'mkpts0_f'):
        assert torch.equal(data[k][:n0], solo[k][:n0]), k
    assert float((data['mkpts1_f'][:n0] - solo['mkpts1_f'][:n0]).abs().max()) < 1e-3
    assert int(solo['solver_status'][0]) == int(data['solver_status'][0]) == 1


def test_precision_modes_deviation(model):
    """The optional half-precision convolution modes against the fp32 path on 4 pairs: 'fp16-fine' must leave every
    coarse decision bit-identical; 'fp16' is reported by match-set IoU (documented deviation, not parity)."""
    import copy
    m = copy.deepcopy(model)
    data32, _, _ = _batch(4, 31)
    with torch.no_grad():
        m.set_precision('fp32')(data32) if False else m(data32)
        d16f, _, _ = _batch(4, 31)
        m.set_precision('fp16-fine')
        m(d16f)
        d16, _, _ = _batch(4, 31)
        m.set_precision('fp16')
        m(d16)
        m.set_precision('fp32')
    assert torch.equal(d16f['feats_c'], data32['feats_c'])
    for k in ['b_ids', 'i_ids', 'j_ids', 'mconf']:
        assert torch.equal(d16f[k], data32[k]), k
    dev = (d16f['mkpts1_f'] - data32['mkpts1_f']).abs()
    assert dev.mean().item() < 0.03 and dev.median().item() < 0.01
    s32 = set(zip(data32['b_ids'].tolist(), data32['i_ids'].tolist(), data32['j_ids'].tolist()))
    s16 = set(zip(d16['b_ids'].tolist(), d16['i_ids'].tolist(), d16['j_ids'].tolist()))
    iou = len(s32 & s16) / len(s32 | s16)
    print('fp16 backbone match-set IoU', iou)
    assert iou > 0.95


@pytest.mark.parametrize('mode', ['mixed16', 'fp16'])
def test_mixed16_mode_deviation(model, mode):
    """LoFTR.set_precision('mixed16') (plain-fp16 backbone K9, bf16 K1, plain-fp16 K2, split-fp16 fused encoder layers) and 'fp16'
    (16-bit operands in the fused kernels too: far_attn_block_f16, far_mlp_fused_f16, plain k|v-state / q-apply projections): the
    16-bit-operand class, reported next to the parity line in bench.py's other_modes -- never the parity configuration.
    Bars: match-set IoU > 0.95 against the fp32-grade path, the EMM feature block within 1e-2 of its scale on the same
    coarse features, and the mode must really switch K1 / K2 (different bits) and switch back (identical bits)."""
    import copy
    m = copy.deepcopy(model)
    d32, _, _ = _batch(4, 31)
    dm, _, _ = _batch(4, 31)
    d32b, _, _ = _batch(4, 31)
    with torch.no_grad():
        m(d32)
        feats = torch.cat([d32['featmap0'], d32['featmap1']], 0).contiguous()      # the coarse tokens (2B, 4800, 256)
        blk32 = m.loftr_regress.emm(feats)
        m.set_precision(mode)
        assert m.coarse_matching.bf16 and m.loftr_regress.emm.cross_attn.plain16 and not m.backbone.trunk_split
        m(dm)
        blk16 = m.loftr_regress.emm(feats)
        m.set_precision('fp32')
        assert not m.coarse_matching.bf16 and not m.loftr_regress.emm.cross_attn.plain16 and m.backbone.trunk_split
        m(d32b)
        blk32b = m.loftr_regress.emm(feats)
    for k in ['b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts1_f']:
        assert torch.equal(d32[k], d32b[k]), k
    assert torch.equal(blk32, blk32b)
    s32 = set(zip(d32['b_ids'].tolist(), d32['i_ids'].tolist(), d32['j_ids'].tolist()))
    s16 = set(zip(dm['b_ids'].tolist(), dm['i_ids'].tolist(), dm['j_ids'].tolist()))
    iou = len(s32 & s16) / len(s32 | s16)
    dev = float((blk16 - blk32).abs().max() / blk32.abs().max())
    print(f'{mode}: match-set IoU {iou:.4f}, EMM block max deviation / scale {dev:.3g}')
    assert iou > 0.95
    assert 0 < dev < 1e-2


def test_precision_by_stage(model):
    """LoFTR.set_precision accepts a set of stage names (round 6).  The stages that cannot touch a match decision -- the FPN's fine
    branch, the fine-level layers, the head's attention -- keep b_ids / i_ids / j_ids / mconf bit-identical to the parity line when
    they run on 16-bit operands; the trunk, the coarse layers and K1 do not.  Names are checked; the removed vendor 'bf16' mode (MIOpen
    under autocast) is no longer a mode of the package."""
    import copy
    m = copy.deepcopy(model)
    d32, _, _ = _batch(2, 33)
    with torch.no_grad():
        m(d32)
        for st, exact in ((('fpn', 'fine_layers', 'k2'), True), (('trunk',), False), (('coarse_dense',), False), (('k1',), False)):
            m.set_precision(st)
            assert m.precision_stages == tuple(s for s in m.STAGES if s in st)
            d, _, _ = _batch(2, 33)
            m(d)
            same = all(d[k].shape == d32[k].shape and torch.equal(d[k], d32[k]) for k in ('b_ids', 'i_ids', 'j_ids', 'mconf'))
            assert same == exact, (st, same)
        m.set_precision('fp32')
        assert m.precision_stages == () and m.backbone.trunk_split and m.backbone.fpn_split
        d, _, _ = _batch(2, 33)
        m(d)
    for k in ('b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts1_f'):
        assert torch.equal(d[k], d32[k]), k
    with pytest.raises(ValueError):
        m.set_precision('bf16')
    with pytest.raises(ValueError):
        m.set_precision(('trunk', 'no_such_stage'))


def test_activation_range_recovery_end_to_end(model):
    """A checkpoint whose activations leave the split-fp16 range of the default exponent (|a| > 4094): the stem's BatchNorm
    scale and shift are multiplied by 2^13, so the first feature map reaches ~3e4 and everything downstream grows with it.
    The reference's fp32 modules do not care; here the forward must notice (device flag), widen the range (activation
    exponent 4 -> lower, unfused fine layers, exact-f32 K1 / K2), re-run, and return finite results that agree with the
    vendor fp32 modules on the same weights -- not inf / NaN, not an empty match list."""
    import copy
    import warnings
    from far_amd import ops
    big = copy.deepcopy(model)
    with torch.no_grad():
        big.backbone.bn1.weight.mul_(2.0 ** 13)
        big.backbone.bn1.bias.mul_(2.0 ** 13)
    data1, _, _ = _batch(1, 5)
    ops.overflow_flag('cuda').zero_()
    with torch.no_grad(), warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        big(data1)
        big.forward_rt_prediction(dict(data1, loftr_rt=torch.eye(3, 4, dtype=torch.float64, device='cuda'),
                                       num_correspondences=torch.tensor([100], device='cuda'),
                                       num_correspondences_before_ransac=torch.tensor([200], device='cuda'),
                                       inliers_best_tight=torch.tensor([50], device='cuda'),
                                       inliers_best_ultra_tight=torch.tensor([5], device='cuda')))
    assert big.act_exp < 4 and any('activation' in str(w.message) for w in rec)
    print('[range recovery] activation exponent after the re-run(s):', big.act_exp)
    assert big.coarse_matching.variant == 'f32' and not big.loftr_fine.layers[0].fused_attn
    for key in ('feats_c', 'featmap_f0', 'featmap0', 'expec_f', 'mkpts1_f', 'mconf'):
        assert torch.isfinite(data1[key]).all(), key
    assert len(data1['b_ids']) > 100
    assert not ops.activation_overflowed('cuda')                     # the re-run left the flag clear
    # the vendor fp32 modules on the same weights (the reference's arithmetic class) are the yardstick
    bb = big.backbone
    with torch.no_grad():
        x = torch.cat([data1['image0'], data1['image1']], 0)
        x1 = bb.layer1(bb.relu(bb.bn1(bb.conv1(x))))
        x2 = bb.layer2(x1)
        x3_out = bb.layer3_outconv(bb.layer3(x2))
        fine = bb._fpn_plain(x1, x2, x3_out)
    deviation('range recovery feats_c', data1['feats_c'], x3_out, atol=1e-4 * float(x3_out.abs().max()))
    deviation('range recovery featmap_f', torch.cat([data1['featmap_f0'], data1['featmap_f1']], 0), fine, atol=1e-4 * float(fine.abs().max()))
    assert float(x3_out.abs().max()) > 4094.0                        # the test does exercise the range
    # a second batch runs at the widened setting without another re-run
    data2, _, _ = _batch(1, 6)
    with torch.no_grad(), warnings.catch_warnings(record=True) as rec2:
        warnings.simplefilter('always')
        big(data2)
    assert not [w for w in rec2 if 'activation' in str(w.message)] and len(data2['b_ids']) > 100


@pytest.mark.parametrize('mode', ['fp16', 'mixed16'])
def test_activation_range_recovery_in_the_16_bit_modes(model, mode):
    """The same oversized checkpoint under the 16-bit-operand modes: plain-fp16 K9 / K2 / K13 / K14 launches raise the same device flag
    (|a| 2^act_exp beyond the fp16 range), the step is re-run at a wider range, the results are finite and the mode's settings
    (bf16 K1, plain K2) are still in place afterwards."""
    import copy
    import warnings
    from far_amd import ops
    from far_amd.pipeline import test_step
    big = copy.deepcopy(model).set_precision(mode)
    with torch.no_grad():
        big.backbone.bn1.weight.mul_(2.0 ** 13)
        big.backbone.bn1.bias.mul_(2.0 ** 13)
    d, _, _ = _batch(1, 5)
    ops.overflow_flag('cuda').zero_()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        test_step(big, d, H=256)
    assert big.act_exp < 4 and any('activation' in str(w.message) for w in rec)
    for key in ('feats_c', 'featmap0', 'mkpts1_f', 'mconf', 'regressed_rt'):
        assert torch.isfinite(d[key]).all(), key
    assert len(d['b_ids']) > 100 and not ops.activation_overflowed('cuda')
    assert big.coarse_matching.bf16 and big.loftr_regress.emm.cross_attn.plain16
    d2, _, _ = _batch(1, 6)
    with warnings.catch_warnings(record=True) as rec2:
        warnings.simplefilter('always')
        test_step(big, d2, H=256)
    assert not [w for w in rec2 if 'activation' in str(w.message)] and torch.isfinite(d2['regressed_rt']).all()


def test_activation_range_guard_ignores_stale_flags_and_keeps_its_state_on_bad_inputs(model):
    """ADVICE r3 (medium).  (1) A flag left set by somebody else -- a backward launch with an inf gradient, another module --
    must not widen this module on its next clean forward: _guarded clears it on entry.  (2) Non-finite INPUTS raise the flag
    at every exponent: the forward must raise ActivationOverflow at once and leave act_exp, the fused fine layers and the K1 /
    K2 variants exactly as they were (round 3 walked down to act_exp = -24 and stayed there)."""
    import copy
    import warnings
    from far_amd import ops
    m = copy.deepcopy(model)
    data1, _, _ = _batch(1, 5)
    state0 = (m.act_exp, m.coarse_matching.variant, m.loftr_fine.layers[0].fused_attn, m.loftr_fine.layers[0].fused_mlp)
    ops.overflow_flag('cuda').fill_(1)                       # stale report
    with torch.no_grad(), warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        m(data1)
    assert not [w for w in rec if 'activation' in str(w.message)]
    assert (m.act_exp, m.coarse_matching.variant, m.loftr_fine.layers[0].fused_attn, m.loftr_fine.layers[0].fused_mlp) == state0
    assert len(data1['b_ids']) > 100
    bad, _, _ = _batch(1, 6)
    bad['image0'] = bad['image0'].clone()
    bad['image0'][0, 0, 96:104, 96:104] = float('inf')        # (a NaN pixel alone can vanish in the stem's ReLU: max(NaN, 0) = 0)
    with torch.no_grad(), warnings.catch_warnings(record=True), pytest.raises(ops.ActivationOverflow):
        m(bad)
    assert (m.act_exp, m.coarse_matching.variant, m.loftr_fine.layers[0].fused_attn, m.loftr_fine.layers[0].fused_mlp) == state0
    # a non-finite WEIGHT: finite inputs, the flag fires at every exponent down to the floor -> the walk is undone, not kept
    m2 = copy.deepcopy(model)
    with torch.no_grad():
        m2.backbone.layer1[0].conv1.weight[3, 5, 1, 1] = float('nan')
    ok_in, _, _ = _batch(1, 6)
    with torch.no_grad(), warnings.catch_warnings(record=True), pytest.raises(ops.ActivationOverflow):
        m2(ok_in)
    assert (m2.act_exp, m2.coarse_matching.variant, m2.loftr_fine.layers[0].fused_attn, m2.loftr_fine.layers[0].fused_mlp) == state0
    good, _, _ = _batch(1, 7)                                # and the module still works, at the default setting
    with torch.no_grad():
        m(good)
    assert len(good['b_ids']) > 100 and torch.isfinite(good['mconf']).all()


def test_train_step_with_padded_masks(model):
    """A padded-mask batch (images of different valid sizes on one 480 x 640 canvas, data['mask0'] / ['mask1'] at the coarse grid:
    coarse_matching.py:28-57, 110-117, loftr.py:103-111) through pipeline.train_step on the GPU: round 3 raised NotImplementedError.
    The coarse stage takes the dense differentiable form (pinned to the reference by golden G18 on CPU tensors); here: no predicted
    or sampled match touches a padded cell, conf_matrix is zero there, the coarse loss weights follow compute_c_weight
    (loftr_loss.py:184-191), every parameter gets a finite gradient, and the batch without masks still takes the sparse kernels."""
    import copy
    from far_amd.config import far_train_config, RunCfg
    from far_amd.losses import LoFTRLoss
    from far_amd.pipeline import train_step
    cfg = far_train_config()
    m = copy.deepcopy(model).train()
    loss_fn = LoFTRLoss(cfg).train()
    base = synth.synth_training_batch(2, seed=78, device='cuda')
    # valid extents at the coarse grid: pair 0: 52 x 80 / 60 x 70; pair 1: 60 x 64 / 48 x 80 (rows x columns); pixels beyond are zero
    ext0, ext1 = [(52, 80), (60, 64)], [(60, 70), (48, 80)]
    m0 = torch.zeros(2, 60, 80, dtype=torch.bool, device='cuda')
    m1 = torch.zeros(2, 60, 80, dtype=torch.bool, device='cuda')
    batch = dict(base)
    batch['image0'], batch['image1'] = base['image0'].clone(), base['image1'].clone()
    for n in range(2):
        m0[n, :ext0[n][0], :ext0[n][1]] = True
        m1[n, :ext1[n][0], :ext1[n][1]] = True
        batch['image0'][n, :, 8 * ext0[n][0]:, :] = 0; batch['image0'][n, :, :, 8 * ext0[n][1]:] = 0
        batch['image1'][n, :, 8 * ext1[n][0]:, :] = 0; batch['image1'][n, :, :, 8 * ext1[n][1]:] = 0
    keep = m0.flatten(1)[base['spv_b_ids'], base['spv_i_ids']] & m1.flatten(1)[base['spv_b_ids'], base['spv_j_ids']]
    for k in ('spv_b_ids', 'spv_i_ids', 'spv_j_ids'):
        batch[k] = base[k][keep]
    batch['mask0'], batch['mask1'] = m0, m1
    torch.manual_seed(5)
    train_step(m, batch, loss_fn, RunCfg('prior_ransac', 2), H=512, seed=0)
    conf = batch['conf_matrix']
    assert conf is not None and conf.shape == (2, 4800, 4800) and conf.requires_grad          # the dense form
    pad0, pad1 = ~m0.flatten(1), ~m1.flatten(1)
    # a padded row against a valid column (and the reverse) is exactly 0; padded against padded is the uniform 1 / (L S) the
    # reference's masked_fill(-1e9) leaves there (both softmaxes see a constant line), far below the match threshold
    c0, c1 = conf.detach()[0], conf.detach()[1]
    assert float(c0[pad0[0]][:, ~pad1[0]].abs().max()) == 0.0 and float(c1[~pad0[1]][:, pad1[1]].abs().max()) == 0.0
    assert float(c0[pad0[0]][:, pad1[0]].max()) < 1e-7
    b, i, j = batch['b_ids'], batch['i_ids'], batch['j_ids']
    assert bool(m0.flatten(1)[b, i].all()) and bool(m1.flatten(1)[b, j].all())
    a0 = torch.tensor([e[0] * e[1] for e in ext0]); a1 = torch.tensor([e[0] * e[1] for e in ext1])
    n_train = int(int(torch.minimum(a0, a1).sum()) * m.coarse_matching.train_coarse_percent)   # compute_max_candidates :46-57
    assert len(b) == n_train, (len(b), n_train)
    assert torch.isfinite(batch['loss']).all()
    batch['loss'].backward()
    for k, p_ in m.named_parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all(), k
    # the same module, a batch without masks: back on the sparse kernels
    plain = dict(base)
    torch.manual_seed(5)
    train_step(m, plain, loss_fn, RunCfg('prior_ransac', 2), H=512, seed=0)
    assert plain['conf_matrix'] is None and plain['conf_pos'].shape == base['spv_b_ids'].shape


def test_training_step_gradients_are_bit_identical_from_run_to_run(model):
    """every reduction of the training step has a fixed order (K16 slabs, K6 / stem partials, the fine-window
    scatter since round 4), so two runs of the same step from the same state give bit-identical losses and parameter gradients.
    Reported per parameter; MIOpen's BatchNorm backward and ATen's interpolation backward are the vendor pieces left in the step."""
    import copy
    from far_amd.config import far_train_config, RunCfg
    from far_amd.losses import LoFTRLoss
    from far_amd.pipeline import train_step
    cfg = far_train_config()
    loss_fn = LoFTRLoss(cfg).train()
    base = synth.synth_training_batch(1, seed=79, device='cuda')
    runs = []
    for _ in range(3):
        m = copy.deepcopy(model).train()
        batch = dict(base)
        torch.manual_seed(11)
        train_step(m, batch, loss_fn, RunCfg('prior_ransac', 2), H=512, seed=0)
        batch['loss'].backward()
        torch.cuda.synchronize()
        runs.append((float(batch['loss']), {k: p_.grad.clone() for k, p_ in m.named_parameters()}))
    assert runs[0][0] == runs[1][0] == runs[2][0]
    differing = [k for k in runs[0][1] if not (torch.equal(runs[0][1][k], runs[1][1][k]) and torch.equal(runs[0][1][k], runs[2][1][k]))]
    groups = sorted(set(k.rsplit('.', 2)[0] for k in differing))
    print(f'[determinism] {len(runs[0][1]) - len(differing)} of {len(runs[0][1])} parameter gradients bit-identical over 3 runs; differing: {groups}')
    assert not differing


def test_step_runs_under_one_activation_range_guard(model):
    """pipeline.test_step wraps matcher -> solver -> head -> solver -> head in ONE guard (LoFTR.guarded_sequence): the calls inside
    do not read the overflow flag (three host synchronisations per step gone), the flag is read once behind the step.  (1) On a
    normal checkpoint the results are bit-identical to the same calls under their own guards.  (2) On a checkpoint whose activations
    leave the default range the WHOLE step is re-run at the widened range: finite results equal to a second (clean) run's, a
    caller-supplied state (no priorRT at entry) restored before the re-run, the flag clear afterwards."""
    import copy
    import warnings
    from far_amd import ops
    from far_amd.config import RunCfg
    from far_amd.pipeline import test_step
    from far_amd.supervision import compute_supervision_RT
    keys = ['b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts1_f', 'loftr_rt', 'regressed_rt']
    d1, _, _ = _batch(2, 41)
    d2 = {k: v for k, v in d1.items()}
    calls = []
    orig = ops.activation_overflowed
    ops.activation_overflowed = lambda dev, reset=True: (calls.append(1), orig(dev, reset))[1]
    try:
        test_step(model, d1, H=256)
        n_seq = len(calls)
        cfg = RunCfg('prior_ransac', 2)
        with torch.no_grad():                                   # the same calls, each under its own guard
            model(d2)
            d2['translation_scale'] = None
            compute_supervision_RT(d2, cfg, H=256, seed=0)
            model.forward_rt_prediction(d2)
            compute_supervision_RT(d2, cfg, H=256, seed=0)
            model.forward_rt_prediction(d2)
        n_each = len(calls) - n_seq
    finally:
        ops.activation_overflowed = orig
    assert n_seq == 1 and n_each == 3, (n_seq, n_each)
    for k in keys:
        assert torch.equal(d1[k], d2[k]), k
    assert np.array_equal(d1['priorRT'], d2['priorRT'])
    # (2) the range recovery through the one guard
    big = copy.deepcopy(model)
    with torch.no_grad():
        big.backbone.bn1.weight.mul_(2.0 ** 13)
        big.backbone.bn1.bias.mul_(2.0 ** 13)
    d3, _, _ = _batch(1, 5)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        test_step(big, d3, H=256)
    assert big.act_exp < 4 and any('activation' in str(w.message) for w in rec)
    assert not ops.activation_overflowed('cuda')
    d4, _, _ = _batch(1, 5)
    with warnings.catch_warnings(record=True) as rec2:
        warnings.simplefilter('always')
        test_step(big, d4, H=256)                               # a clean run at the widened setting
    assert not [w for w in rec2 if 'activation' in str(w.message)]
    for k in keys:
        assert torch.isfinite(d3[k].float()).all() and torch.equal(d3[k], d4[k]), k
    assert len(d3['b_ids']) > 100


def test_guard_treats_an_exception_behind_an_overflow_as_the_overflow(model):
    """with one guard per step the solver rounds and the head run on the matcher's inf / NaN outputs before the flag is read;
    an exception they raise on that garbage (zero matches, non-finite keypoints) must widen the range and re-run like the plain
    overflow does -- and an exception WITHOUT the flag is the caller's and propagates unchanged."""
    import copy
    import warnings
    from far_amd import ops
    m = copy.deepcopy(model)
    dev = torch.device('cuda', torch.cuda.current_device())
    runs = []

    def fn():
        runs.append(m.act_exp)
        if len(runs) == 1:
            ops.overflow_flag(dev).fill_(1)                 # what an out-of-range launch leaves behind ...
            raise ValueError('degenerate match set')        # ... and what a later stage makes of its outputs
        return 'clean'
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        assert m.guarded_sequence(fn, dev) == 'clean'
    assert runs == [4, 0] and m.act_exp == 0 and any('activation' in str(w.message) for w in rec)
    assert not ops.activation_overflowed(dev)

    def plain():
        raise KeyError('not an overflow')
    with pytest.raises(KeyError):
        m.guarded_sequence(plain, dev)
    assert m.act_exp == 0
