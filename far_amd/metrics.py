"""Pose-error metrics of the evaluation step that follows the hot path.

Mirrors mp3d_loftr/src/utils/metrics.py: relative_pose_error :17-36 (batched here), error_auc :307-324,
the Matterport summary of aggregate_metrics :359-376.  Per-pair errors are a handful of flops: batched torch on
the device the poses live on (float64), aggregation on the host like the reference.
"""
import numpy as np
import torch


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
        out[f'auc@{thr}'] = float(np.trapz(y, x) / thr) if hasattr(np, 'trapz') else float(np.trapezoid(y, x) / thr)
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


def aggregate_pose_metrics(t_errs, R_errs, t_errs_abs, successful_fits=None):
    """The pose part of aggregate_metrics (metrics.py:343-376)."""
    t_errs, R_errs, t_abs = (np.asarray(a, np.float64) for a in (t_errs, R_errs, t_errs_abs))
    res = {
        'tr rot mean err': np.round(np.mean(t_errs), 2), 'tr rot median err': np.round(np.median(t_errs), 2),
        'tr rot pct < 30': np.round(100 * np.mean(t_errs < 30), 1),
        'tr abs mean err': np.round(np.mean(t_abs), 2), 'tr abs median err': np.round(np.median(t_abs), 2),
        'tr abs pct < 1': np.round(100 * np.mean(t_abs < 1), 1),
        'rot mean err': np.round(np.mean(R_errs), 2), 'rot median err': np.round(np.median(R_errs), 2),
        'rot pct < 30': np.round(100 * np.mean(R_errs < 30), 1), 'dset size': len(t_errs),
    }
    if successful_fits is not None:
        res['pct successful fits'] = np.round(100 * np.mean(np.asarray(successful_fits, np.float64)), 1)
    res.update(error_auc(np.maximum(R_errs, t_errs)))
    return res
