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


def test_eval_pairs_tool_walks_a_pair_list_and_prints_the_reference_table(tmp_path):
    """tools/eval_pairs.py: a pair list with ground-truth poses -> pipeline.test_step in batches ->
    far_amd.metrics.aggregate_metrics (pinned to the reference's own function by golden G16, tests/test_oracle_golden.py) -> the lines
    test_epoch_end prints (lightning_loftr.py:483-492), for both minimal solvers.  Synthetic pairs (ground truth R = I, t = -x),
    6 pairs in batches of 4 + 2: the per-pair errors the tool accumulates equal a direct evaluation of each pair alone, the table
    is aggregate_metrics of them, and the command line prints the reference's keys."""
    import subprocess
    import sys
    from far_amd import metrics as fm
    from far_amd.config import RunCfg
    from far_amd.loftr import LoFTR
    from far_amd.pipeline import test_step
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import eval_pairs
    n = 6
    im0, im1 = synth.synth_image_pair(n, seed=21)
    T = np.tile(np.eye(4), (n, 1, 1))
    T[:, 0, 3] = -1.0
    p = str(tmp_path / 'pairs.npz')
    np.savez(p, image0=(im0[:, 0] * 255).round().astype(np.uint8), image1=(im1[:, 0] * 255).round().astype(np.uint8), K0=synth.MP3D_K,
             K1=synth.MP3D_K, T_0to1=T, identifiers=np.array([f'scene_{i}' for i in range(n)]))
    pairs = eval_pairs.load_pairs(p)
    assert pairs['image0'].shape == (n, 1, 480, 640) and pairs['K0'].shape == (n, 3, 3) and float(pairs['image0'].max()) <= 1.0
    m = LoFTR(far_eval_config()).eval()
    synth.load_synthetic(m, seed=0)
    m = m.cuda()
    far, sol, acc = eval_pairs.evaluate(m, pairs, minimal=8, batch=4, hyp=512)
    assert far['dset size'] == n and sol['dset size'] == n and acc['identifiers'] == [f'scene_{i}' for i in range(n)]
    assert set(far) >= {'rot median err', 'rot mean err', 'tr rot median err', 'tr abs median err', 'pct successful fits', 'auc@5', 'auc@10', 'auc@20',
                        'prec@5e-04'}
    assert far == fm.aggregate_metrics(acc)
    # pair 5 alone (batch 32 = 32 x batch 1 bit for bit, and the sampling hash is keyed by the pair's index in its batch: pair 5 is
    # index 1 of the second batch -> seed ^ 1 reproduces its samples at index 0)
    d = {'image0': torch.from_numpy(pairs['image0'][5:6]).cuda(), 'image1': torch.from_numpy(pairs['image1'][5:6]).cuda(),
         'K0': torch.from_numpy(pairs['K0'][5:6]).float().cuda(), 'K1': torch.from_numpy(pairs['K1'][5:6]).float().cuda(),
         'T_0to1': torch.from_numpy(T[5:6]).cuda(), 'dataset_name': ['mp3d']}
    cfg = RunCfg('prior_ransac', 2, minimal_solver=8)
    test_step(m, d, run_cfg=cfg, H=512, seed=0 ^ 1)
    fm.compute_pose_errors(d, cfg, H=512, seed=0 ^ 1)
    assert abs(d['R_errs'][0] - acc['R_errs'][5]) < 1e-9 and abs(d['t_errs'][0] - acc['t_errs'][5]) < 1e-9
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'eval_pairs.py'), '--pairs', p, '--batch', '4', '--hyp', '512',
                          '--out', str(tmp_path / 't.json')], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    for key in ('minimal solver 8: FAR pose', 'minimal solver 5: solver pose alone', 'rot median err', 'tr rot median err', 'auc@20', 'dset size 6'):
        assert key in out.stdout, (key, out.stdout)
    tab = json.load(open(tmp_path / 't.json'))
    assert tab['minimal_8']['far']['rot median err'] == float(far['rot median err']) and 'minimal_5' in tab
