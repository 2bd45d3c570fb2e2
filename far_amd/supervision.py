"""Solver call site of the pipeline: spvs_RT / compute_supervision_RT with the reference's data-dict contract
(mp3d_loftr/src/loftr/utils/supervision.py:184-240), batched over all pairs in one solver launch.

Reference quirk: its per-pair loop keeps only the LAST pair's pose and counts.
Here every pair's result is kept, stacked along dim 0; with one pair the tensors have the reference's shapes:
loftr_rt / expec_rt (3, 4) float64, expec_e (3, 3), count tensors (1,).
"""
import torch

from . import ops
from .solver import estimate_pose_batch


def spvs_RT(data, config, H=2048, seed=0):
    pixel_thr = config.TRAINER.RANSAC_PIXEL_THR
    solver = config.LOFTR.SOLVER
    K0, K1 = data['K0'], data['K1']
    B = K0.shape[0]
    dev = data['mkpts0_f'].device
    prior = data['priorRT'] if (solver == 'prior_ransac' and 'priorRT' in data) else None     # :198-201
    if 'match_counts' in data:
        counts = [int(c) for c in data['match_counts']]
    else:
        counts = torch.bincount(data['m_bids'], minlength=B).cpu().tolist()
    out = estimate_pose_batch(data['mkpts0_f'], data['mkpts1_f'], counts, K0, K1, pixel_thr, solver, prior,
                              H=H, seed=seed)
    # K11: [R | t] with the identity fallback (:218-224), E, and the count tensors (zero below 5 correspondences,
    # metrics.py:83-85) in one launch
    rt, E, before, after, tight, ultra = ops.pose_pack(out, out['offsets'])
    data.update({
        'loftr_rt': rt[0] if B == 1 else rt,
        'expec_rt': rt[0] if B == 1 else rt,
        'expec_e': E[0] if B == 1 else E,
        'num_correspondences_before_ransac': before,
        'num_correspondences_after_ransac': after,
        'num_correspondences': after,
        'inliers_best_tight': tight,
        'inliers_best_ultra_tight': ultra,
        'solver_inlier_mask': out['mask'],
        'solver_status': out['status'],
    })


def compute_supervision_RT(data, config, **kw):
    src = data['dataset_name'][0] if 'dataset_name' in data else 'mp3d'
    if src.lower() in ['mp3d', 'interiornet_streetlearn']:
        spvs_RT(data, config, **kw)
    else:
        raise NotImplementedError(src)
