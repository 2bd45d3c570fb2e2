"""The evaluation step that follows the hot path: PL_LoFTR._compute_metrics
(mp3d_loftr/src/lightning/lightning_loftr.py:227-264) needs

    compute_symmetrical_epipolar_errors(data)       src/utils/metrics.py:58-77    -> data['epi_errs']
    compute_pose_errors(data, config)               :198-303   -> R_errs / t_errs / t_errs_abs / inliers / successful_fits /
                                                                  pred_R / pred_t / num_correspondences_*_ransac
    aggregate_metrics(metrics, epi_err_thr)         :339-377   (error_auc :307-324, epidist_prec :326-337)

with the reference's data-dict keys.  Per-pair quantities are a handful of flops: batched torch ops on the device the
matches / poses live on (float64, no per-pair Python loop over tensors), one host read per call for the lists the
reference's callers index; aggregation on the host like the reference.  The solver branch of compute_pose_errors runs
K4 once for the whole batch (far_amd.solver.estimate_pose_batch).
"""
from collections import OrderedDict

import numpy as np
import torch

from .pose6d import pose_mean_6d, pose_std_6d, rotation_6d_to_matrix


def relative_pose_error_batch(T_0to1, R, t, ignore_gt_t_thr=0.0):
    """T_0to1 (B,4,4) or (B,3,4) ground truth; R (B,3,3), t (B,3) estimates (any float dtype/device).
    Returns t_err_deg, R_err_deg, t_err_abs, each (B,) float64 -- metrics.py:17-36 for every pair at once."""
    T = T_0to1.to(torch.float64)
    R = R.to(device=T.device, dtype=torch.float64)
    t = t.to(device=T.device, dtype=torch.float64)
    t_gt = T[:, :3, 3]
    n = t.norm(dim=1) * t_gt.norm(dim=1)
    cos_t = torch.clamp((t * t_gt).sum(1) / n, -1.0, 1.0)
    t_err = torch.rad2deg(torch.acos(cos_t))
    t_err = torch.minimum(t_err, 180 - t_err)                               # E sign ambiguity (:24)
    t_err = torch.where(t_gt.norm(dim=1) < ignore_gt_t_thr, torch.zeros_like(t_err), t_err)
    t_abs = (t - t_gt).norm(dim=1)                                          # :29
    R_gt = T[:, :3, :3]
    cos_r = ((R.transpose(1, 2) @ R_gt).diagonal(dim1=1, dim2=2).sum(1) - 1) / 2   # :33
    R_err = torch.rad2deg(torch.acos(torch.clamp(cos_r, -1.0, 1.0)).abs())
    return t_err, R_err, t_abs


def error_auc(errors, thresholds=(5, 10, 20)):
    """metrics.py:307-324 (note: the reference ignores its `thresholds` argument and always uses 5/10/20)."""
    errors = [0] + sorted(list(np.asarray(errors, dtype=np.float64)))
    recall = list(np.linspace(0, 1, len(errors)))
    out = {}
    for thr in (5, 10, 20):
        last = int(np.searchsorted(errors, thr))
        y = recall[:last] + [recall[last - 1]]
        x = errors[:last] + [thr]
        out[f'auc@{thr}'] = float((np.trapezoid if hasattr(np, 'trapezoid') else np.trapz)(y, x) / thr)
    return out


def error_auc_device(errors, thresholds=(5, 10, 20)):
    """error_auc (metrics.py:307-324) on the device the errors live on: sort + trapezoid rule as torch ops, one small
    host read of the three numbers at the end (the per-pair errors never leave the GPU)."""
    e = torch.sort(errors.reshape(-1).to(torch.float64))[0]
    n = e.numel()
    x = torch.cat([torch.zeros(1, dtype=torch.float64, device=e.device), e])
    y = torch.linspace(0, 1, n + 1, dtype=torch.float64, device=e.device)
    out = {}
    for thr in (5, 10, 20):
        last = int(torch.searchsorted(x, torch.tensor(float(thr), dtype=torch.float64, device=e.device)))
        xs = torch.cat([x[:last], torch.tensor([float(thr)], dtype=torch.float64, device=e.device)])
        ys = torch.cat([y[:last], y[last - 1:last]])
        out[f'auc@{thr}'] = float(torch.trapezoid(ys, xs) / thr)
    return out


def symmetric_epipolar_distance(pts0, pts1, E, K0, K1):
    """metrics.py:39-56 for one pair: pts (M, 2) pixels, E (3, 3), K (3, 3) -> (M,) squared symmetric epipolar distance."""
    b = torch.zeros(pts0.shape[0], dtype=torch.int64, device=pts0.device)
    return _sym_epi(pts0, pts1, b, E[None], K0[None], K1[None])


def _sym_epi(pts0, pts1, bids, E, K0, K1):
    """Every match of every pair at once: E, K0, K1 are (B, 3, 3), bids (M,) selects the pair of each match."""
    dt = torch.promote_types(pts0.dtype, E.dtype)
    E, K0, K1 = (x.to(device=pts0.device, dtype=dt) for x in (E, K0, K1))
    k0, k1 = K0[bids], K1[bids]
    p0 = (pts0.to(dt) - k0[:, :2, 2]) / torch.stack([k0[:, 0, 0], k0[:, 1, 1]], 1)            # :46-47
    p1 = (pts1.to(dt) - k1[:, :2, 2]) / torch.stack([k1[:, 0, 0], k1[:, 1, 1]], 1)
    one = torch.ones_like(p0[:, :1])
    p0, p1 = torch.cat([p0, one], 1), torch.cat([p1, one], 1)                                   # :48-49
    Em = E[bids]
    Ep0 = torch.einsum('mij,mj->mi', Em, p0)                                                   # pts0 @ E.T (:51)
    p1Ep0 = (p1 * Ep0).sum(-1)                                                                 # :52
    Etp1 = torch.einsum('mj,mji->mi', p1, Em)                                                  # pts1 @ E (:53)
    return p1Ep0 ** 2 * (1.0 / (Ep0[:, 0] ** 2 + Ep0[:, 1] ** 2) + 1.0 / (Etp1[:, 0] ** 2 + Etp1[:, 1] ** 2))   # :55


def cross_product_matrix(t):
    """kornia's numeric.cross_product_matrix for (B, 3) vectors (metrics.py:63)."""
    z = torch.zeros_like(t[:, 0])
    return torch.stack([z, -t[:, 2], t[:, 1], t[:, 2], z, -t[:, 0], -t[:, 1], t[:, 0], z], 1).view(-1, 3, 3)


def compute_symmetrical_epipolar_errors(data):
    """metrics.py:58-77.  Updates data['epi_errs'] (M,): the errors of pair 0's matches, then pair 1's, ... -- the
    concatenation the reference builds; with m_bids sorted (every evaluation path) that is the matches' own order."""
    T = data['T_0to1']
    E = cross_product_matrix(T[:, :3, 3]) @ T[:, :3, :3]                                       # :63-64
    m_bids = data['m_bids']
    p0, p1 = data['mkpts0_f'].detach(), data['mkpts1_f'].detach()
    if m_bids.numel() > 1 and bool((m_bids[1:] < m_bids[:-1]).any()):                          # training-time order
        order = torch.sort(m_bids, stable=True)[1]
        m_bids, p0, p1 = m_bids[order], p0[order], p1[order]
    data.update({'epi_errs': _sym_epi(p0, p1, m_bids, E, data['K0'], data['K1'])})


def failed_fit_translation(drew_point_cloud=False):
    """The translation the reference reports for a failed fit (metrics.py:253, :284): np.random.rand(3) - .5 from the
    generator seeded with 0 at :243; the prior branches of estimate_pose draw their 300 x 3 point cloud first (:103, :130)."""
    rs = np.random.RandomState(0)
    if drew_point_cloud:
        rs.uniform(low=-3.0, high=3.0, size=(300, 3))
    return rs.rand(3) - .5


def compute_pose_errors(data, config, estimate_pose_fn=None, H=2048, seed=0):
    """metrics.py:198-303 with the reference's keys: R_errs, t_errs, t_errs_abs (lists of float), inliers (list: the
    bool mask of a fit, zeros for a failed one, 0 on the head branch), successful_fits, num_correspondences_before /
    after_ransac (lists, successful fits only), pred_R / pred_t (numpy, the LAST pair's, as the reference leaves them).

    Branches as the reference: 'regressed_rt' in data -> the head's pose (:228-233; the reference reads row 0 for every
    pair, i.e. batch size 1 -- here pair b reads row b, identical at B = 1); else matches in data -> the solver
    (:234-258), here ONE K4 launch for all pairs (estimate_pose_fn: a per-pair callable with the reference's
    estimate_pose signature instead, used to pin the bookkeeping against the reference); else the identity (:283-287).
    `config.SAVE_PREDS` set: data['correspondences'] (+ '_feats' with SAVE_HARD_CORRES) of the last pair (:262-282; the
    'ground_truth' depth lookup dereferences `depths = None` in the reference and is not reproduced)."""
    from .solver import _branch, estimate_pose_batch
    pixel_thr = config.TRAINER.RANSAC_PIXEL_THR
    conf = config.TRAINER.RANSAC_CONF
    solver = config.LOFTR.SOLVER
    K0, K1 = data['K0'], data['K1']
    B = K0.shape[0]
    T = data['T_0to1']
    data.update({k: [] for k in ('R_errs', 't_errs', 't_errs_abs', 'inliers', 'successful_fits', 'pred_R', 'pred_t',
                                 'num_correspondences_before_ransac', 'num_correspondences_after_ransac')})
    prior = None
    if solver == 'prior_ransac' and 'priorRT' in data:                                         # :217-222
        prior = data['priorRT']
        prior = prior.cpu().numpy() if torch.is_tensor(prior) else np.asarray(prior)
    dev = T.device
    if 'regressed_rt' in data:                                                                 # :228-233
        p = data['regressed_rt'].detach().float()
        if p.shape[0] != B:
            p = p[:1].expand(B, -1)
        v = p * pose_std_6d.to(p.device) + pose_mean_6d.to(p.device)
        R_all, t_all = rotation_6d_to_matrix(v[:, 3:]), v[:, :3]
        inliers, fits = [0] * B, [0] * B
    elif 'mkpts0_c' in data or 'mkpts0_f' in data:                                             # :234-258
        m_bids, pts0, pts1 = data['m_bids'], data['mkpts0_f'], data['mkpts1_f']
        dev = pts0.device
        R_all = torch.eye(3, dtype=torch.float64, device=dev).repeat(B, 1, 1)
        t_all = torch.zeros(B, 3, dtype=torch.float64, device=dev)
        inliers, fits = [None] * B, [0] * B
        counts = torch.bincount(m_bids, minlength=B).cpu().tolist() if m_bids.numel() else [0] * B
        order = None
        if B > 1 and m_bids.numel() and bool((m_bids[1:] < m_bids[:-1]).any()):
            order = torch.sort(m_bids, stable=True)[1]
            pts0, pts1 = pts0[order], pts1[order]
        offs = np.concatenate([[0], np.cumsum(counts)])
        failed_mask = np.zeros(int(m_bids.numel()))       # :253: np.zeros(mask.shape[0]) -- as long as ALL pairs' matches
        scale = data.get('translation_scale')
        if estimate_pose_fn is not None:
            for b in range(B):
                pr = None if prior is None else (prior if prior.ndim == 2 else prior[b])
                ret, n_after, _, _ = estimate_pose_fn(pts0[offs[b]:offs[b + 1]], pts1[offs[b]:offs[b + 1]], K0[b], K1[b], pixel_thr,
                                                      conf=conf, translation_scale=scale, solver=solver, priorRT=pr)
                if ret is None:
                    t_all[b] = torch.from_numpy(failed_fit_translation()).to(dev)
                    inliers[b] = failed_mask
                else:
                    R_all[b], t_all[b] = torch.as_tensor(ret[0]).to(dev), torch.as_tensor(ret[1]).to(dev)
                    inliers[b], fits[b] = np.asarray(ret[2]), 1
                    data['num_correspondences_before_ransac'].append(counts[b])
                    data['num_correspondences_after_ransac'].append(n_after)
        else:
            pr = None if prior is None else np.array(np.broadcast_to(prior.reshape(-1, 3, 4), (B, 3, 4)))
            out = estimate_pose_batch(pts0, pts1, counts, K0, K1, pixel_thr, solver, pr, H=H, seed=seed,
                                      minimal=getattr(config.LOFTR, 'MINIMAL_SOLVER', 8))
            status = out['status'].cpu().numpy().astype(bool)
            n_after = out['num_after'].cpu()
            mask = out['mask'].cpu().numpy() > 0
            drew = _branch(solver, pr is not None) != 'ransac'
            ok = torch.from_numpy(status).to(dev)
            t = out['t'] if scale is None else out['t'] * scale.to(dev)
            R_all = torch.where(ok[:, None, None], out['R'], R_all)
            t_fail = torch.from_numpy(np.stack([failed_fit_translation(drew and counts[b] >= 5) for b in range(B)])).to(dev)
            t_all = torch.where(ok[:, None], t, t_fail)
            for b in range(B):
                fits[b] = int(status[b])
                inliers[b] = mask[offs[b]:offs[b + 1]] if status[b] else failed_mask
                if status[b]:
                    data['num_correspondences_before_ransac'].append(counts[b])
                    data['num_correspondences_after_ransac'].append(n_after[b])
        save = getattr(config, 'SAVE_PREDS', None)
        if save is not None and B:                                                             # :262-282, last pair
            b = B - 1
            c0, c1 = pts0[offs[b]:offs[b + 1]], pts1[offs[b]:offs[b + 1]]
            inl = torch.from_numpy(np.asarray(inliers[b]).astype(bool)).to(dev)
            if 'ground_truth' in save or getattr(config, 'SAVE_HARD_CORRES', False):
                if 'ground_truth' in save:
                    raise NotImplementedError("SAVE_PREDS 'ground_truth': the reference dereferences depths = None here (metrics.py:241, :271)")
                if getattr(config, 'SAVE_CORR_AFTER_RANSAC', False):
                    c0, c1 = c0[inl], c1[inl]
                feats = torch.stack([data['featmap0'], data['featmap1']], dim=1).reshape(1, 2, 60, 80, 256)
                data['correspondences_feats'] = compute_correspondences_feats(c0, c1, feats)
                data['correspondences'] = torch.cat([c0, c1], dim=-1).reshape(-1, 2, 2)
            else:
                data['correspondences'] = torch.cat([c0, c1, inl.float().unsqueeze(1)], dim=-1).cpu()
    else:                                                                                      # :283-287
        R_all = torch.eye(3, dtype=torch.float64, device=dev).repeat(B, 1, 1)
        t_all = torch.from_numpy(failed_fit_translation()).to(dev).repeat(B, 1)
        inliers, fits = [0] * B, [0] * B
    te, Re, ta = relative_pose_error_batch(T.to(R_all.device), R_all, t_all)                   # :289
    host = torch.stack([te, Re, ta]).cpu().numpy()                                             # the step's one host read
    data['t_errs'], data['R_errs'], data['t_errs_abs'] = (list(host[0]), list(host[1]), list(host[2]))
    data['inliers'], data['successful_fits'] = inliers, fits
    if B:
        data['pred_R'] = R_all[-1].cpu().numpy()                                               # :298-299: the last pair's
        data['pred_t'] = t_all[-1].cpu().numpy()


def compute_correspondences_feats(kpts0, kpts1, feats):
    """metrics.py:176-183: coarse features (1, 2, 60, 80, 256) at the correspondences' cells -> (N, 2, 256)."""
    def cells(k):
        return torch.clamp((k / 8)[:, 0].long(), 0, 79), torch.clamp((k / 8)[:, 1].long(), 0, 59)
    x0, y0 = cells(kpts0)
    x1, y1 = cells(kpts1)
    return torch.stack([feats[0, 0, y0, x0], feats[0, 1, y1, x1]], dim=1)


def epidist_prec(errors, thresholds, ret_dict=False):
    """metrics.py:326-337: mean over pairs of the fraction of a pair's matches with epipolar error below thr."""
    precs = []
    for thr in thresholds:
        per_pair = [np.mean(np.asarray(e) < thr) if len(e) > 0 else 0 for e in errors]
        precs.append(np.mean(per_pair) if len(per_pair) > 0 else 0)
    if ret_dict:
        return {f'prec@{t:.0e}': p for t, p in zip(thresholds, precs)}
    return precs


def aggregate_metrics(metrics, epi_err_thr=5e-4):
    """metrics.py:339-377: pose AUC and matching precision over the de-duplicated pairs (DistributedSampler pads the
    last batch: OrderedDict keeps each identifier's LAST index at its first position), the Matterport summary over all
    entries.  `metrics`: the dict of lists PL_LoFTR gathers (identifiers, epi_errs, R_errs, t_errs, t_errs_abs,
    successful_fits)."""
    unq = list(OrderedDict((iden, i) for i, iden in enumerate(metrics['identifiers'])).values())
    pose_errors = np.max(np.stack([metrics['R_errs'], metrics['t_errs']]), axis=0)[unq]
    aucs = error_auc(pose_errors)
    epi = metrics['epi_errs']
    precs = epidist_prec([epi[u] for u in unq], [epi_err_thr], True)
    res = aggregate_pose_metrics(metrics['t_errs'], metrics['R_errs'], metrics['t_errs_abs'], metrics['successful_fits'], auc=False)
    return {**res, **aucs, **precs}


def aggregate_pose_metrics(t_errs, R_errs, t_errs_abs, successful_fits=None, auc=True):
    """The pose part of aggregate_metrics (metrics.py:343-376), in the reference's key order."""
    t_errs, R_errs, t_abs = (np.asarray(a, np.float64) for a in (t_errs, R_errs, t_errs_abs))
    res = {
        'tr rot mean err': np.round(np.mean(t_errs), 2), 'tr rot median err': np.round(np.median(t_errs), 2),
        'tr rot pct < 30': np.round(100 * np.mean(t_errs < 30), 1),
        'tr abs mean err': np.round(np.mean(t_abs), 2), 'tr abs median err': np.round(np.median(t_abs), 2),
        'tr abs pct < 1': np.round(100 * np.mean(t_abs < 1), 1),
        'rot mean err': np.round(np.mean(R_errs), 2), 'rot median err': np.round(np.median(R_errs), 2),
        'rot pct < 30': np.round(100 * np.mean(R_errs < 30), 1),
    }
    if successful_fits is not None:
        res['pct successful fits'] = np.round(100 * np.mean(np.asarray(successful_fits, np.float64)), 1)
    res['dset size'] = len(t_errs)
    if auc:
        res.update(error_auc(np.maximum(R_errs, t_errs)))
    return res
