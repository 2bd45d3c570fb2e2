"""BASELINE configs[4]'s resolution (544x720: coarse grid 68x90, L = S = 6120; fine grid 272x360) end to end on the GPU:
LoFTR.forward + compute_supervision_RT, against golden G11 (the reference's own matcher at that size,
tools/make_goldens.py:g11_matcher_544x720) and against the oracle.  The EMM head is tied to the 60x80 grid in the
reference (`pos_embed (1, 4800, 256)`, transformer.py:194, :319-326): matcher + solver are this configuration's path."""
import json
import os

import numpy as np
import pytest
import torch

from far_amd import synth
from far_amd.config import RunCfg, far_eval_config
from tests.util import deviation

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
HW = (544, 720)
K_C5 = np.array([[590.0, 0, 360.], [0, 590.0, 272.], [0, 0, 1.]])


@pytest.fixture(scope='module')
def model():
    from far_amd.loftr import LoFTR
    m = LoFTR(far_eval_config()).eval()
    synth.load_synthetic(m, seed=0)
    return m.cuda()


def _batch(N, seed):
    im0, im1 = synth.synth_image_pair(N, seed=seed, hw=HW)
    K = torch.from_numpy(np.stack([K_C5] * N)).cuda()
    return {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(),
            'dataset_name': ['mp3d']}, im0, im1


def test_matcher_544x720_vs_reference_golden(model):
    g = np.load(os.path.join(G, 'g11_matcher_544x720.npz'))
    data, _, _ = _batch(1, 5)
    with torch.no_grad():
        model(data)
    assert tuple(data['hw0_c']) == (68, 90) and tuple(data['hw0_f']) == (272, 360) and data['featmap0'].shape == (1, 6120, 256)
    # bars = ~3x the deviations measured on MI355X (printed by deviation(); fp32-grade kernels vs the fp32 reference)
    deviation('c5 feats_c', data['feats_c'][:, ::16, ::7, ::9], g['feats_c_sample'], atol=6e-5, rtol=1e-4)
    deviation('c5 featmap_f0', data['featmap_f0'][:, ::16, ::31, ::37], g['featmap_f0_sample'], atol=1.2e-4, rtol=1e-4)
    deviation('c5 featmap0 (tokens)', data['featmap0'][0, ::97], g['featmap0_sample'], atol=7e-5, rtol=1e-4)
    gi, gj = data['i_ids'].cpu().numpy(), data['j_ids'].cpu().numpy()
    got = dict(zip(gi.tolist(), gj.tolist()))
    ref = dict(zip(g['i_ids'].tolist(), g['j_ids'].tolist()))
    safe = (np.abs(g['rowmax'] - 0.2) > 1e-4) & (g['rowgap'] > 1e-4)            # margin 1e-4
    assert safe.sum() > 0.85 * 6120
    for i in np.nonzero(safe)[0]:
        assert (i in got) == (i in ref), i
        if i in got:
            assert got[i] == ref[i]
    common = [i for i in ref if i in got]
    assert len(common) > 0.99 * len(ref) > 2000
    a = np.array([{i: n for n, i in enumerate(gi.tolist())}[i] for i in common])
    b = np.array([{i: n for n, i in enumerate(g['i_ids'].tolist())}[i] for i in common])
    deviation('c5 mconf', data['mconf'][a], g['mconf'][b], atol=1.5e-4, rtol=0)
    deviation('c5 mkpts1_f', data['mkpts1_f'][a], g['mkpts1_f'][b], atol=4e-3, rtol=0)
    deviation('c5 expec_f', data['expec_f'][a], g['expec_f'][b], atol=1.5e-3, rtol=0)


def test_c5_two_pairs_matcher_and_solver_vs_oracle(model):
    """2 pairs at 544x720: matcher stage set-wise against the oracle's own matcher, then both solver branches
    (plain RANSAC = first round of 'prior_ransac' without a prior; prior RANSAC with a pose prior) bit-exact against
    the oracle on the GPU's own correspondences."""
    from far_amd.supervision import compute_supervision_RT
    from oracle import model as om
    from oracle import solver as osv
    cfg = far_eval_config()
    data, im0, im1 = _batch(2, 31)
    man = json.load(open(os.path.join(G, 'g8_state_dict_manifest.json')))
    w = om.Weights(synth.synthetic_state_dict({k: tuple(v) for k, v in man.items()}))
    with torch.no_grad():
        model(data)
    odata = om.matcher_forward(w, cfg, im0, im1)
    gm = set(zip(data['b_ids'].tolist(), data['i_ids'].tolist(), data['j_ids'].tolist()))
    rm = set(zip(odata['b_ids'].tolist(), odata['i_ids'].tolist(), odata['j_ids'].tolist()))
    assert len(gm & rm) > 0.99 * len(rm) > 3000, (len(gm & rm), len(rm))
    deviation('c5 featmap0 vs oracle', data['featmap0'], odata['featmap0'], atol=1e-4, rtol=1e-4)
    mk0, mk1 = data['mkpts0_f'].cpu().numpy(), data['mkpts1_f'].cpu().numpy()
    bids = data['m_bids'].cpu().numpy()
    pcl = np.random.RandomState(0).uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
    Hn, seed = 512, 9
    prior = None
    for rnd in range(2):
        if rnd == 1:       # a pose prior as forward_rt_prediction would export it (numpy (B, 3, 4), loftr.py:188-192)
            prior = np.stack([np.concatenate([np.eye(3), np.array([[-1.0], [0.02 * b], [0.05]])], 1) for b in range(2)])
            data['priorRT'] = prior
        compute_supervision_RT(data, RunCfg('prior_ransac'), H=Hn, seed=seed)
        rt = data['loftr_rt'].cpu().numpy()
        mask = data['solver_inlier_mask'].cpu().numpy().astype(bool)
        for b in range(2):
            sel = bids == b
            ret, na, ti, ul, _ = osv.estimate_pose(mk0[sel], mk1[sel], K_C5, K_C5, 0.5, solver='prior_ransac',
                                                   priorRT=None if prior is None else prior[b], seed=seed, pair=b,
                                                   H=Hn, pcl=pcl)
            assert ret is not None and int(data['solver_status'][b]) == 1
            R, t, m, E = ret
            assert np.linalg.norm(rt[b] - np.concatenate([R, t[:, None]], 1)) < 1e-4          # north_star: 1e-4 Frobenius
            np.testing.assert_array_equal(mask[sel], m)                                         # bit-exact inlier mask
            assert (int(data['num_correspondences'][b]), int(data['inliers_best_tight'][b]),
                    int(data['inliers_best_ultra_tight'][b])) == (na, ti, ul)
            assert int(data['num_correspondences_before_ransac'][b]) == sel.sum()


def test_head_refuses_the_544x720_grid(model):
    """The reference head cannot run on a 68x90 grid either (pos_embed / positional table are 60x80): loud error."""
    data, _, _ = _batch(1, 5)
    with torch.no_grad():
        model(data)
        data.update({'loftr_rt': torch.eye(3, 4, dtype=torch.float64).cuda(), 'num_correspondences': torch.tensor([10]).cuda(),
                     'num_correspondences_before_ransac': torch.tensor([10]).cuda(),
                     'inliers_best_tight': torch.tensor([1]).cuda(), 'inliers_best_ultra_tight': torch.tensor([0]).cuda()})
        with pytest.raises((ValueError, RuntimeError)):
            model.forward_rt_prediction(data)
