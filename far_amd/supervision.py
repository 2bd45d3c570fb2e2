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
    mk0, mk1, m_bids = data['mkpts0_f'], data['mkpts1_f'], data['m_bids']
    M = int(mk0.shape[0])
    if mk1.shape[0] != M or m_bids.shape[0] != M:
        raise ValueError(f'spvs_RT: mkpts0_f / mkpts1_f / m_bids disagree on the match count '
                         f'({M}, {mk1.shape[0]}, {m_bids.shape[0]})')
    # The reference selects each pair's correspondences with `mask = m_bids == bs` (:209-210): order-independent.
    # The batched solver wants them as contiguous per-pair segments.  Inference emits matches ordered by (b, i) and
    # K1 hands over the per-pair counts (`match_counts`, host ints); in training `m_bids` is UNSORTED for B > 1
    # (the sampled prediction / padded ground-truth indices of coarse_matching.py:216-240), and a caller may have
    # filtered the matches: then the segments are built here by a stable sort of m_bids, and the inlier mask is
    # scattered back to the caller's order.
    order = None
    counts = [int(c) for c in data['match_counts']] if 'match_counts' in data else None
    if counts is not None and (len(counts) != B or sum(counts) != M or data.get('b_ids') is not m_bids):
        counts = None          # stale (the matches were filtered / re-ordered after K1 wrote it): do not trust it
    if counts is None:
        counts = torch.bincount(m_bids, minlength=B).cpu().tolist() if M else [0] * B
        if len(counts) != B:
            raise ValueError(f'spvs_RT: m_bids refers to pair {len(counts) - 1} but K0 holds {B} pairs')
        if B > 1 and M and bool((m_bids[1:] < m_bids[:-1]).any()):
            order = torch.sort(m_bids, stable=True)[1]
            mk0, mk1 = mk0[order], mk1[order]
    assert sum(counts) == M
    out = estimate_pose_batch(mk0, mk1, counts, K0, K1, pixel_thr, solver, prior, H=H, seed=seed)
    if order is not None:                                  # the mask back in the order the caller's matches are in
        mask = torch.empty_like(out['mask'])
        mask[order] = out['mask']
        out['mask'] = mask
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
