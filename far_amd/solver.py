"""Pose solver front end on kernel K4: `estimate_pose` with the reference's signature and return contract
(mp3d_loftr/src/utils/metrics.py:80-174), plus the batched form the harness uses.

Dispatch mirrors the reference:
  solver == 'prior_ransac' and priorRT is not None -> prior RANSAC (inl_th 3e-7, biased sampling, prior score)
  solver == 'prior_ransac_noprior'                 -> same machinery without prior
  otherwise ('ransac', or 'prior_ransac' w/o prior) -> plain RANSAC with the cv2 threshold semantics
The minimal solver is the normalized 8-point on the GPU in every branch by default (the reference executes OpenCV's
5-point on the host; see DESIGN.md "K4" for what parity means here); pairs with 5..7 correspondences -- and every pair
with minimal=5 -- get hypotheses from Nister's five-point solver (far_amd/csrc/solver5_f64.inc), which is not degenerate
on planar scenes.
"""
import numpy as np
import torch

from . import ops

_PCL_CACHE = {}


def prior_point_cloud(device):
    """The 300-point cloud of metrics.py:103: np.random.uniform(-3, 3, (300, 3)) drawn right after the
    np.random.seed(0) that supervision.py:207 / metrics.py:243 issue before every solver call."""
    key = str(device)
    if key not in _PCL_CACHE:
        rs = np.random.RandomState(0)
        pcl = rs.uniform(low=-3.0, high=3.0, size=(300, 3)).astype(np.float32)
        _PCL_CACHE[key] = torch.from_numpy(pcl).to(device)
    return _PCL_CACHE[key]


def _branch(solver, has_prior):
    if solver == 'prior_ransac' and has_prior:
        return 'prior'
    if solver == 'prior_ransac_noprior':
        return 'noprior'
    return 'ransac'


def estimate_pose_batch(kpts0, kpts1, counts, K0, K1, thresh, solver='ransac', priorRT=None, H=2048, seed=0, minimal=8, cache=None):
    """kpts0/kpts1: (Mtot, 2) fp32 GPU, concatenated per pair in order; counts: per-pair M (host ints);
    K0/K1: (B, 3, 3); priorRT: None or (B, 3, 4) numpy/tensor.  Returns the dict of ops.solve_pose_batch.
    cache: a dict living as long as the batch (the data dict's): the float64 intrinsics and the inlier thresholds of one solver round are
    reused by the next (they are a handful of tiny launches each, in a stretch of the step where the host is the bottleneck)."""
    dev = kpts0.device
    counts = [int(c) for c in counts]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    mode = _branch(solver, priorRT is not None)
    many = mode != 'ransac'
    tens = torch.is_tensor(K0) and torch.is_tensor(K1)
    # the cache entry holds the K0 / K1 tensor objects themselves (so their addresses cannot be handed to another tensor while the
    # entry lives) and their versions: a replaced batch['K0'] or an in-place rescale of the intrinsics misses
    kkey = ('K', tuple(K0.shape), str(dev)) if (cache is not None and tens) else None
    hit = cache.get(kkey) if kkey is not None else None
    stamp = (ops.tensor_version(K0), ops.tensor_version(K1)) if kkey is not None else None
    if hit is not None and hit[0] is K0 and hit[1] is K1 and hit[2] == stamp:
        K0d, K1d = hit[3], hit[4]
    else:
        K0d = K0.to(device=dev, dtype=torch.float64)
        K1d = K1.to(device=dev, dtype=torch.float64)
        if kkey is not None:
            cache[kkey] = (K0, K1, stamp, K0d, K1d)
            for k in [k for k in cache if isinstance(k, tuple) and k and k[0] == 'th']:      # thresholds derived from the old intrinsics
                del cache[k]
    tkey = ('th', kkey, float(thresh), many, len(counts)) if kkey is not None else None
    if tkey is not None and tkey in cache:
        inl_th = cache[tkey]
    else:
        if many:
            inl_th = torch.full((len(counts),), 3e-7, dtype=torch.float64, device=dev)            # metrics.py:117
        else:
            # metrics.py:94 -- note the reference averages K0[0,0], K1[1,1] twice
            f = (K0d[:, 0, 0] + K1d[:, 1, 1] + K0d[:, 0, 0] + K1d[:, 1, 1]) / 4
            inl_th = (thresh / f) ** 2
        inl_th = inl_th.contiguous()
        if tkey is not None:
            cache[tkey] = inl_th
    prior = pcl = None
    if mode == 'prior':
        if torch.is_tensor(priorRT) and priorRT.is_cuda:          # the head's prior, still on the device (no round trip through the host)
            prior = priorRT.to(device=dev, dtype=torch.float32).reshape(-1, 3, 4).contiguous()
        else:
            prior = torch.as_tensor(np.asarray(priorRT), dtype=torch.float32).reshape(-1, 3, 4).pin_memory().to(dev, non_blocking=True)
        pcl = prior_point_cloud(dev)
    return ops.solve_pose_batch(kpts0.float().contiguous(), kpts1.float().contiguous(), offs, K0d, K1d,
                                inl_th.contiguous(), many, priorRT=prior, pcl=pcl, prior_lambda=0.3, H=H, seed=seed,
                                minimal=minimal)


def estimate_pose(kpts0, kpts1, K0, K1, thresh, conf=0.99999, translation_scale=None, solver='ransac',
                  priorRT=None, H=2048, seed=0, minimal=8):
    """Single-pair form with the reference's return contract:
    (ret, num_correspondences_after_ransac, inliers_best_tight, inliers_best_ultra_tight) where
    ret is None or (R: f64 (3,3) on device, t: f64 (3,), mask: np.bool (M,), E: cpu f64 (3,3))."""
    if len(kpts0) < 5:                                                                        # :83-85
        return None, 0, 0, 0
    out = estimate_pose_batch(kpts0, kpts1, [len(kpts0)], K0[None], K1[None], thresh, solver,
                              None if priorRT is None else np.asarray(priorRT)[None], H=H, seed=seed, minimal=minimal)
    host = {k: out[k].cpu() for k in ['status', 'num_after', 'tight', 'ultra']}
    num_after = int(host['num_after'][0])
    tight, ultra = int(host['tight'][0]), int(host['ultra'][0])
    if not int(host['status'][0]):
        return None, num_after, tight, ultra
    t = out['t'][0]
    if translation_scale is not None:
        t = t * translation_scale.to(t.device)
    ret = (out['R'][0], t, out['mask'].cpu().numpy() > 0, out['E'][0].cpu())
    return ret, torch.tensor(num_after), tight, ultra
