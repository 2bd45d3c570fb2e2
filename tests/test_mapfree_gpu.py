"""K12 (far_corr_volume_warp_f32): the Map-free 6DReg correlation-volume warp on the GPU against the reference's own
module (golden G13) and against the float64 oracle, full 92 x 68 grid included."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _full_inputs(g):
    B, H, W = (int(v) for v in g['f_shape'])
    rng = np.random.default_rng(33)
    rng.standard_normal((2, 32, 12, 9)); rng.standard_normal((2, 32, 12, 9))
    amp = float(g['f_amp'])
    v0 = (amp * rng.standard_normal((B, 32, H, W))).astype(np.float32)
    v1 = (amp * rng.standard_normal((B, 32, H, W))).astype(np.float32)
    v1[:, :, : H // 2] = 2.0 * v0[:, :, : H // 2][:, :, ::-1] + 0.3 * v1[:, :, : H // 2]
    return v0, v1


def test_corr_volume_warp_vs_reference_golden_and_oracle():
    from far_amd import ops
    from oracle import mapfree as omf
    g = np.load(os.path.join(G, 'g13_mapfree_cvw.npz'))
    agg = ops.corr_volume_warp(torch.from_numpy(g['s_vol0']).cuda(), torch.from_numpy(g['s_vol1']).cuda()).cpu().numpy()
    assert agg.shape == (2, 67, 12, 9)
    np.testing.assert_array_equal(agg[:, :32], g['s_vol0'])                       # the vol0 block of the concatenation
    np.testing.assert_allclose(agg, g['s_agg'], rtol=2e-5, atol=2e-6)             # vs the reference (fp32)
    np.testing.assert_allclose(agg, omf.corr_volume_warp(g['s_vol0'], g['s_vol1']), rtol=3e-5, atol=4e-6)    # fp32 vs float64 (measured 2.8e-6)
    v0, v1 = _full_inputs(g)
    agg = ops.corr_volume_warp(torch.from_numpy(v0).cuda(), torch.from_numpy(v1).cuda()).cpu().numpy()
    ref64 = omf.corr_volume_warp(v0, v1)
    d = np.abs(agg - ref64)
    print(f'[k12] 92x68: max |agg - float64 oracle| = {d.max():.2e} (warped features), {d[:, 64:66].max():.2e} (grid), '
          f'{d[:, 66].max():.2e} (max score)')
    np.testing.assert_allclose(agg, ref64, rtol=3e-5, atol=4e-6)
    np.testing.assert_allclose(agg[:, :, ::7, ::5], g['f_agg_sample'], rtol=5e-5, atol=5e-6)    # vs the reference run
    np.testing.assert_allclose(agg[:, 66], g['f_max_score'], rtol=5e-5, atol=1e-7)


def test_corr_volume_warp_batch_and_ragged_grid():
    """Several pairs, a grid whose size is no multiple of the 32-column tile or the 128-row block."""
    from far_amd import ops
    from oracle import mapfree as omf
    rng = np.random.default_rng(5)
    v0 = (0.5 * rng.standard_normal((3, 32, 23, 17))).astype(np.float32)
    v1 = (0.5 * rng.standard_normal((3, 32, 23, 17))).astype(np.float32)
    agg = ops.corr_volume_warp(torch.from_numpy(v0).cuda(), torch.from_numpy(v1).cuda()).cpu().numpy()
    np.testing.assert_allclose(agg, omf.corr_volume_warp(v0, v1), rtol=3e-5, atol=4e-6)
    with pytest.raises(Exception):
        ops.corr_volume_warp(torch.zeros(1, 16, 4, 4).cuda(), torch.zeros(1, 16, 4, 4).cuda())     # D must be 32
