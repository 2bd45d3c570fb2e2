"""CPU tests of host-side logic that carries no kernel: solver dispatch, config, fail-loud behaviour."""
import numpy as np
import pytest
import torch

from far_amd import _lib, ops, solver
from far_amd.config import RunCfg, far_eval_config


def test_solver_branch_mirrors_reference_dispatch():
    # metrics.py:100, :130, :153
    assert solver._branch('prior_ransac', True) == 'prior'
    assert solver._branch('prior_ransac', False) == 'ransac'       # falls through to the cv2.RANSAC branch
    assert solver._branch('prior_ransac_noprior', False) == 'noprior'
    assert solver._branch('ransac', False) == 'ransac'


def test_prior_point_cloud_is_the_reference_draw():
    # np.random.seed(0) (supervision.py:207) then np.random.uniform(-3, 3, (300, 3)) (metrics.py:103)
    np.random.seed(0)
    ref = np.random.uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
    np.testing.assert_array_equal(solver.prior_point_cloud('cpu').numpy(), ref)


def test_fewer_than_five_keypoints_returns_none():
    k = torch.zeros(4, 2)
    assert solver.estimate_pose(k, k, torch.eye(3), torch.eye(3), 0.5) == (None, 0, 0, 0)     # metrics.py:83-85


def test_ops_fail_loudly_without_gpu_tensors():
    x = torch.zeros(1, 32, 32)
    with pytest.raises(_lib.FarHipError):
        ops.dual_softmax_stats(x, x)
    with pytest.raises(_lib.FarHipError):
        ops.linear_attention(x, x, x, 1)


def test_config_matches_far_eval_setting():
    c = far_eval_config()
    assert c['coarse']['layer_names'] == ['self', 'cross'] * 3 and c['solver'] == 'prior_ransac'
    assert c['match_coarse']['thr'] == 0.2 and c['match_coarse']['border_rm'] == 2
    r = RunCfg()
    assert r.TRAINER.RANSAC_PIXEL_THR == 0.5 and r.LOFTR.SOLVER == 'prior_ransac'


def test_inference_on_cpu_raises_instead_of_falling_back():
    # no grad, eval mode, CPU tensors: the kernel path is the only path and it needs the GPU
    from far_amd.loftr import CoarseMatching
    m = CoarseMatching(far_eval_config()['match_coarse']).eval()
    with torch.no_grad(), pytest.raises(_lib.FarHipError):
        m(torch.zeros(1, 4, 32), torch.zeros(1, 4, 32), {'hw0_c': (2, 2), 'hw1_c': (2, 2), 'hw0_i': (16, 16)})


def test_cached_prediction_format_roundtrip(tmp_path):
    from far_amd import cache_io
    g = torch.Generator().manual_seed(0)
    B = 3
    data = {'loftr_rt': torch.randn(B, 3, 4, generator=g, dtype=torch.float64), 'num_correspondences': torch.tensor([5, 700, 0]),
            'featmap0': torch.randn(B, 4800, 256, generator=g), 'featmap1': torch.randn(B, 4800, 256, generator=g)}
    cache_io.save_batch(str(tmp_path), 'test', [7, 8, 11], data)
    # the reference's reader side: one .pt per quantity per pair id (interiornet_streetlearn.py:108-118)
    one = torch.load(str(tmp_path / 'test' / 'coarse_features' / '8.pt'))
    assert one.shape == (2, 4800, 256) and torch.equal(one[1], data['featmap1'][1])
    assert torch.load(str(tmp_path / 'test' / 'loftr_preds' / '11.pt')).shape == (3, 4)
    back = cache_io.load_batch(str(tmp_path), 'test', [11, 7])
    assert torch.equal(back['loftr_rt'], data['loftr_rt'][[2, 0]]) and back['num_correspondences'].tolist() == [0, 5]
    assert torch.equal(back['featmap0'], data['featmap0'][[2, 0]]) and back['inliers_best_tight'].tolist() == [0, 0]
