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
    dev_prior = data.get('_priorRT_device')          # (the numpy array it mirrors, the same values as a device tensor): forward_rt_prediction
    if prior is not None and dev_prior is not None and dev_prior[0] is prior:
        prior = dev_prior[1]                          # the caller did not replace data['priorRT']: skip the upload
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
    out = estimate_pose_batch(mk0, mk1, counts, K0, K1, pixel_thr, solver, prior, H=H, seed=seed,
                              minimal=getattr(config.LOFTR, 'MINIMAL_SOLVER', 8), cache=data.setdefault('_solver_cache', {}))
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


# =====================================================================================================================
# Coarse ground-truth supervision from depth (training only): spvs_coarse of src/loftr/utils/supervision.py:34-137 with
# warp_kpts of src/loftr/utils/geometry.py:5-56.  Same outputs for the sparse training path -- spv_b_ids / spv_i_ids /
# spv_j_ids, spv_w_pt0_i, spv_pt1_i -- batched (no per-sample Python loops), and WITHOUT conf_matrix_gt: the reference
# scatters the ids into a zero (N, hw0, hw1) fp32 matrix (92 MB per pair) only for its dense loss to find them again
# with `conf_gt == 1` (loftr_loss.py:63); far_amd.losses reads the ids.  dense_gt=True builds the matrix for drop-in use
# of the reference's own loss.  Label generation: a few small torch ops per batch, no kernel of its own.
# =====================================================================================================================
@torch.no_grad()
def warp_kpts(kpts0, depth0, depth1, T_0to1, K0, K1):
    """geometry.py:5-56.  kpts0 (N, L, 2) <x, y>; depth (N, H, W); T (N, 3|4, 4); K (N, 3, 3) -> valid (N, L), warped (N, L, 2)."""
    N, L = kpts0.shape[:2]
    bidx = torch.arange(N, device=kpts0.device)[:, None].expand(N, L)
    k0 = kpts0.round().long()
    d0 = depth0[bidx, k0[..., 1], k0[..., 0]]                                                  # :22-25
    nonzero = d0 != 0
    kh = torch.cat([kpts0, torch.ones_like(kpts0[:, :, :1])], dim=-1) * d0[..., None]          # :31
    cam = K0.inverse() @ kh.transpose(2, 1)                                                    # :32
    wcam = T_0to1[:, :3, :3] @ cam + T_0to1[:, :3, 3:4]                                        # :35
    wdepth = wcam[:, 2, :]
    wh = (K1 @ wcam).transpose(2, 1)                                                           # :39
    wk = wh[:, :, :2] / (wh[:, :, 2:3] + 1e-4)                                                 # :40
    h, w = depth1.shape[1:3]
    covis = (wk[..., 0] > 0) & (wk[..., 0] < w - 1) & (wk[..., 1] > 0) & (wk[..., 1] < h - 1)   # :44-45
    wl = wk.long()
    wl[~covis] = 0
    d1 = depth1[bidx, wl[..., 1], wl[..., 0]]                                                  # :49-51
    consistent = ((d1 - wdepth) / d1).abs() < 0.2                                              # :52
    return nonzero & covis & consistent, wk


@torch.no_grad()
def spvs_coarse(data, config, dense_gt=False):
    """supervision.py:34-137.  config: the dict-like with ['LOFTR']['RESOLUTION'] (or an int coarse scale)."""
    dev = data['image0'].device
    N, _, H0, W0 = data['image0'].shape
    _, _, H1, W1 = data['image1'].shape
    scale = config if isinstance(config, int) else config['LOFTR']['RESOLUTION'][0]
    scale0 = scale * data['scale0'][:, None] if 'scale0' in data else scale                    # :62-63
    scale1 = scale * data['scale1'][:, None] if 'scale0' in data else scale
    h0, w0, h1, w1 = (x // scale for x in (H0, W0, H1, W1))

    def grid(h, w):                       # kornia create_meshgrid(h, w, False): (x, y) pixel coordinates, x fastest
        ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                                torch.arange(w, device=dev, dtype=torch.float32), indexing='ij')
        return torch.stack([xs, ys], -1).reshape(1, h * w, 2).repeat(N, 1, 1)
    pt0_i = scale0 * grid(h0, w0)                                                              # :69-72
    pt1_i = scale1 * grid(h1, w1)
    if 'mask0' in data:                                                                        # :75-77
        pt0_i = pt0_i * data['mask0'].reshape(N, -1, 1).bool()
        pt1_i = pt1_i * data['mask1'].reshape(N, -1, 1).bool()
    _, w_pt0_i = warp_kpts(pt0_i, data['depth0'], data['depth1'], data['T_0to1'], data['K0'], data['K1'])   # :83
    _, w_pt1_i = warp_kpts(pt1_i, data['depth1'], data['depth0'], data['T_1to0'], data['K1'], data['K0'])   # :84
    r0 = (w_pt0_i / scale1).round().long()                                                     # :86-90
    r1 = (w_pt1_i / scale0).round().long()
    near1 = r0[..., 0] + r0[..., 1] * w1
    near0 = r1[..., 0] + r1[..., 1] * w0
    oob = lambda pt, w, h: (pt[..., 0] < 0) | (pt[..., 0] >= w) | (pt[..., 1] < 0) | (pt[..., 1] >= h)      # :96-99
    near1[oob(r0, w1, h1)] = 0
    near0[oob(r1, w0, h0)] = 0
    loop_back = torch.gather(near0, 1, near1)                                                  # :101
    correct = loop_back == torch.arange(h0 * w0, device=dev)[None]
    correct[:, 0] = False                                                                      # :103
    b_ids, i_ids = torch.where(correct)                                                        # :106
    j_ids = near1[b_ids, i_ids]
    if dense_gt:                                                                               # :115-119
        gt = torch.zeros(N, h0 * w0, h1 * w1, device=dev)
        gt[b_ids, i_ids, j_ids] = 1
        data['conf_matrix_gt'] = gt
    n_gt = len(b_ids)
    if n_gt == 0:                                                                              # :122-128
        b_ids = i_ids = j_ids = torch.zeros(1, dtype=torch.long, device=dev)
    # spv_gt_count: what `conf_matrix_gt.any()` tells the reference's loss (loftr_loss.py:65-70) now that no dense matrix exists
    data.update({'spv_b_ids': b_ids, 'spv_i_ids': i_ids, 'spv_j_ids': j_ids, 'spv_w_pt0_i': w_pt0_i, 'spv_pt1_i': pt1_i,
                 'spv_gt_count': n_gt})


def compute_supervision_coarse(data, config, **kw):
    src = data['dataset_name'][0]
    if src.lower() in ['scannet', 'megadepth', 'mp3d']:                                        # :139-145
        spvs_coarse(data, config, **kw)
    else:
        raise ValueError(f'Unknown data source: {src}')


@torch.no_grad()
def spvs_fine(data, config):
    """supervision.py:142-166: the fine-level target of every (sampled) coarse match, in units of the window radius."""
    w_pt0_i, pt1_i = data['spv_w_pt0_i'], data['spv_pt1_i']
    scale = config['LOFTR']['RESOLUTION'][1]
    radius = config['LOFTR']['FINE_WINDOW_SIZE'] // 2
    b, i, j = data['b_ids'], data['i_ids'], data['j_ids']
    scale = scale * data['scale1'][b] if 'scale0' in data else scale                           # :158
    data['expec_f_gt'] = (w_pt0_i[b, i] - pt1_i[b, j]) / scale / radius                        # :161


def compute_supervision_fine(data, config):
    if data['dataset_name'][0].lower() in ['mp3d']:                                            # :170-174
        spvs_fine(data, config)
    else:
        raise NotImplementedError
