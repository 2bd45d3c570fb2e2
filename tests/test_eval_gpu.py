"""SURVEY section 8(f) rows 1-2 on the GPU: pose-error metrics next to the path (vs golden G9 from the reference's
src/utils/metrics.py) and the cached-prediction path (BASELINE configs[3]) at batch 256."""
import json
import os

import numpy as np
import pytest
import torch

from far_amd import synth
from far_amd.config import far_eval_config
from tests.util import two_view_scene

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_pose_error_metrics_on_device_vs_reference_golden():
    """relative_pose_error (metrics.py:17-36) for all pairs at once and error_auc (:307-324), evaluated on the GPU."""
    from far_amd import metrics as fm
    g = np.load(os.path.join(G, 'g9_metrics.npz'))
    T, R, t = (torch.from_numpy(g[k]).cuda() for k in ('T', 'R', 't'))
    te, Re, ta = fm.relative_pose_error_batch(T, R, t)
    assert te.is_cuda and te.dtype == torch.float64
    got = torch.stack([te, Re, ta], 1).cpu().numpy()
    np.testing.assert_allclose(got, g['errs'], rtol=1e-7, atol=2e-5)
    auc = fm.error_auc_device(torch.maximum(te, Re))
    np.testing.assert_allclose([auc["auc@5"], auc["auc@10"], auc["auc@20"]], g["auc"], rtol=1e-6, atol=1e-7)      # (errors agree to 2e-5 deg: acos)
    # float32 poses straight from the solver / head are accepted as well
    te32, Re32, _ = fm.relative_pose_error_batch(T.float(), R.float(), t.float())
    # (float32 inputs: acos near 1 amplifies the 6e-8 input rounding to ~0.02 degrees)
    assert float((te32 - te).abs().max()) < 5e-2 and float((Re32 - Re).abs().max()) < 5e-2


def _cached_model():
    from far_amd.loftr import LoFTR
    cfg = far_eval_config()
    cfg['from_saved_preds'] = 'loftr_preds'
    m = LoFTR(cfg).eval()
    man = json.load(open(os.path.join(G, 'g8_state_dict_manifest.json')))
    sd = synth.synthetic_state_dict({k: tuple(v) for k, v in man.items() if k.startswith('loftr_regress.')})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.cuda()


def test_cached_path_batch_256_head_vs_reference_golden(tmp_path):
    """BASELINE configs[3]: 256 cached pairs through cache_io.load_batch -> head in ONE batch (the reference loads pair
    by pair at batch size 1).  Every pair carries golden G4's features and solver pose, so each of the 256 rows must
    reproduce the reference's regressed_rt (1e-3, north_star's bar on regression outputs)."""
    from far_amd import cache_io
    m = _cached_model()
    g = np.load(os.path.join(G, 'g4_head.npz'))
    rng = np.random.default_rng(14)
    f0 = rng.standard_normal((1, 4800, 256)).astype(np.float32)
    f1 = (0.5 * f0 + rng.standard_normal((1, 4800, 256))).astype(np.float32)
    n = g['counts']
    one = {'loftr_rt': torch.from_numpy(g['loftr_rt'])[None], 'num_correspondences': torch.tensor([int(n[0])]),
           'featmap0': torch.from_numpy(f0), 'featmap1': torch.from_numpy(f1)}
    cache_io.save_batch(str(tmp_path), 'test', [0], one)
    for d in ('loftr_preds', 'loftr_num_correspondences', 'coarse_features'):      # 256 pair ids, one payload
        for i in range(1, 256):
            os.symlink(str(tmp_path / 'test' / d / '0.pt'), str(tmp_path / 'test' / d / f'{i}.pt'))
    B = 256
    data = cache_io.load_batch(str(tmp_path), 'test', list(range(B)), device='cuda')
    data.update({'num_correspondences_before_ransac': torch.full((B,), int(n[1])).cuda(),
                 'inliers_best_tight': torch.full((B,), int(n[2])).cuda(),
                 'inliers_best_ultra_tight': torch.full((B,), int(n[3])).cuda()})
    with torch.no_grad():
        m.forward_rt_prediction(data)
    reg = data['regressed_rt'].cpu().numpy()
    assert reg.shape == (B, 9) and np.asarray(data['priorRT']).shape == (B, 3, 4)
    np.testing.assert_allclose(reg, np.repeat(g['regressed_rt'], B, 0), atol=1e-3 * np.abs(g['regressed_rt']).max(), rtol=1e-3)
    assert float(np.abs(reg - reg[0:1]).max()) < 1e-5 * np.abs(reg).max()          # rows are independent problems


def test_cached_step_solver_plus_head_on_cached_correspondences(tmp_path):
    """cached_step: GPU solver on cached fine correspondences (two-view scenes with known pose) + head, 8 pairs through
    the on-disk format; the solver stage bit-exact against the oracle, the round trip through the files lossless."""
    from far_amd import cache_io
    from far_amd.pipeline import cached_step
    from oracle import solver as osv
    m = _cached_model()
    B = 8
    scenes = [two_view_scene(300 + 150 * b, seed=70 + b) for b in range(B)]
    rng = np.random.default_rng(1)
    feats = torch.from_numpy(rng.standard_normal((2, B, 4800, 256)).astype(np.float32))
    src = {'loftr_rt': torch.eye(3, 4, dtype=torch.float64).repeat(B, 1, 1), 'num_correspondences': torch.zeros(B, dtype=torch.int64),
           'featmap0': feats[0], 'featmap1': feats[1],
           'mkpts0_f': torch.from_numpy(np.concatenate([s[0] for s in scenes])),
           'mkpts1_f': torch.from_numpy(np.concatenate([s[1] for s in scenes])),
           'm_bids': torch.from_numpy(np.concatenate([np.full(len(s[0]), b) for b, s in enumerate(scenes)]))}
    ids = [100 + b for b in range(B)]
    cache_io.save_batch(str(tmp_path), 'val', ids, src)
    data = cache_io.load_batch(str(tmp_path), 'val', ids, device='cuda', correspondences=True)
    assert torch.equal(data['mkpts1_f'].cpu(), src['mkpts1_f']) and data['match_counts'].tolist() == [len(s[0]) for s in scenes]
    K = torch.from_numpy(np.stack([s[2] for s in scenes])).cuda()
    data.update({'K0': K, 'K1': K.clone(), 'dataset_name': ['interiornet_streetlearn']})
    Hn, seed = 512, 4
    from far_amd.config import RunCfg
    cached_step(m, data, RunCfg('prior_ransac', 1), H=Hn, seed=seed)            # one round: the no-prior branch, checkable exactly
    rt = data['loftr_rt'].cpu().numpy()
    for b, (k0, k1, Kb, Rgt, tgt) in enumerate(scenes):
        ret, na, ti, ul, _ = osv.estimate_pose(k0, k1, Kb, Kb, 0.5, solver='prior_ransac', priorRT=None, seed=seed, pair=b, H=Hn)
        assert np.linalg.norm(rt[b] - np.concatenate([ret[0], ret[1][:, None]], 1)) < 1e-8
        assert int(data['num_correspondences'][b]) == na
        assert np.linalg.norm(rt[b][:, :3] - Rgt) < 0.05
    assert data['regressed_rt'].shape == (B, 9) and bool(torch.isfinite(data['regressed_rt']).all())
    d2 = cache_io.load_batch(str(tmp_path), 'val', ids, device='cuda', correspondences=True)
    d2.update({'K0': K, 'K1': K.clone(), 'dataset_name': ['interiornet_streetlearn']})
    cached_step(m, d2, RunCfg('prior_ransac', 2), H=Hn, seed=seed)              # the full two-round schedule runs
    assert d2['loftr_rt'].shape == (B, 3, 4) and bool(torch.isfinite(d2['regressed_rt']).all())
