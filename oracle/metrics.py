"""Oracle for the pose-error metrics (test infrastructure only).  Follows mp3d_loftr/src/utils/metrics.py:17-36."""
import numpy as np


def relative_pose_error(T_0to1, R, t, ignore_gt_t_thr=0.0):
    t_gt = T_0to1[:3, 3]
    n = np.linalg.norm(t) * np.linalg.norm(t_gt)
    t_err = np.rad2deg(np.arccos(np.clip(np.dot(t, t_gt) / n, -1.0, 1.0)))
    t_err = np.minimum(t_err, 180 - t_err)
    if np.linalg.norm(t_gt) < ignore_gt_t_thr:
        t_err = 0
    t_err_abs = np.linalg.norm(t - t_gt)
    R_gt = T_0to1[:3, :3]
    cos = np.clip((np.trace(np.dot(R.T, R_gt)) - 1) / 2, -1., 1.)
    return t_err, np.rad2deg(np.abs(np.arccos(cos))), t_err_abs
