"""Oracle for the evaluation metrics that follow the path (test infrastructure only: imported by tests/ and by nothing
in far_amd/).  numpy float64 restatement of mp3d_loftr/src/utils/metrics.py: relative_pose_error :17-36,
symmetric_epipolar_distance :39-56, compute_symmetrical_epipolar_errors :58-77, the bookkeeping of compute_pose_errors
:198-303, error_auc :307-324, epidist_prec :326-337, aggregate_metrics :339-377.  Pinned by goldens G9 and G16, which
tools/make_goldens.py produced by running the reference's own functions."""
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


def symmetric_epipolar_distance(pts0, pts1, E, K0, K1):
    """metrics.py:39-56: squared symmetric epipolar distance of pixel correspondences under E (normalised by K)."""
    p0 = (pts0 - K0[[0, 1], [2, 2]][None]) / K0[[0, 1], [0, 1]][None]
    p1 = (pts1 - K1[[0, 1], [2, 2]][None]) / K1[[0, 1], [0, 1]][None]
    p0 = np.concatenate([p0, np.ones_like(p0[:, :1])], 1)
    p1 = np.concatenate([p1, np.ones_like(p1[:, :1])], 1)
    Ep0 = p0 @ E.T
    p1Ep0 = np.sum(p1 * Ep0, -1)
    Etp1 = p1 @ E
    return p1Ep0 ** 2 * (1.0 / (Ep0[:, 0] ** 2 + Ep0[:, 1] ** 2) + 1.0 / (Etp1[:, 0] ** 2 + Etp1[:, 1] ** 2))


def cross_matrix(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], t.dtype)


def compute_symmetrical_epipolar_errors(T_0to1, m_bids, mkpts0, mkpts1, K0, K1):
    """metrics.py:58-77: E = [t]x R from the ground truth of each pair, errors concatenated pair after pair."""
    out = []
    for b in range(len(T_0to1)):
        E = cross_matrix(T_0to1[b, :3, 3]) @ T_0to1[b, :3, :3]
        m = m_bids == b
        out.append(symmetric_epipolar_distance(mkpts0[m], mkpts1[m], E, K0[b], K1[b]))
    return np.concatenate(out) if out else np.zeros(0)


def rotation_6d_to_matrix(d6):
    """loftr_loss.py:10-29."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = a1 / np.linalg.norm(a1, axis=-1, keepdims=True)
    b2 = a2 - (b1 * a2).sum(-1, keepdims=True) * b1
    b2 = b2 / np.linalg.norm(b2, axis=-1, keepdims=True)
    return np.stack([b1, b2, np.cross(b1, b2)], -2)


POSE_MEAN_6D = np.array([-0.34898765, 0.17085525, -0.87944315, 0.50275223, 0.03533648, -0.18179045,
                         -0.03533648, 0.98189617, 0.09313615])
POSE_STD_6D = np.array([1.94014405, 0.36770130, 1.88317520, 0.51837117, 0.12717603, 0.65426397,
                        0.12717603, 0.0188729, 0.09709263])


def failed_fit_translation(drew_point_cloud=False):
    """The `np.random.rand(3) - .5` of metrics.py:253 / :284: the generator was seeded with 0 at :243; the prior branches of
    estimate_pose draw their 300x3 point cloud first (:103, :130)."""
    rs = np.random.RandomState(0)
    if drew_point_cloud:
        rs.uniform(low=-3.0, high=3.0, size=(300, 3))
    return rs.rand(3) - .5


def compute_pose_errors(T_0to1, regressed_rt=None, fits=None):
    """metrics.py:198-303, float64.  regressed_rt (B, 9): the head branch (:228-233; the reference reads row 0 for every
    pair -- B = 1 semantics; here row b).  fits: per pair None (failed fit) or (R, t, mask) (:234-258).  Neither: the
    no-correspondence branch (:283-287).  Returns dict of lists with the reference's keys."""
    B = len(T_0to1)
    out = {k: [] for k in ('R_errs', 't_errs', 't_errs_abs', 'inliers', 'successful_fits')}
    for b in range(B):
        if regressed_rt is not None:
            v = regressed_rt[b].astype(np.float64) * POSE_STD_6D + POSE_MEAN_6D
            R, t, inl = rotation_6d_to_matrix(v[3:]), v[:3], 0
            out['successful_fits'].append(0)
        elif fits is not None:
            if fits[b] is None:
                R, t, inl = np.eye(3), failed_fit_translation(), None
                out['successful_fits'].append(0)
            else:
                R, t, inl = fits[b]
                out['successful_fits'].append(1)
        else:
            R, t, inl = np.eye(3), failed_fit_translation(), 0
            out['successful_fits'].append(0)
        te, Re, ta = relative_pose_error(T_0to1[b].astype(np.float64), R, t)
        out['R_errs'].append(Re); out['t_errs'].append(te); out['t_errs_abs'].append(ta); out['inliers'].append(inl)
    return out


def error_auc(errors):
    """metrics.py:307-324."""
    errors = [0] + sorted(list(errors))
    recall = list(np.linspace(0, 1, len(errors)))
    aucs = {}
    for thr in (5, 10, 20):
        last = np.searchsorted(errors, thr)
        y = recall[:last] + [recall[last - 1]]
        x = errors[:last] + [thr]
        aucs[f'auc@{thr}'] = np.trapezoid(y, x) / thr
    return aucs


def epidist_prec(errors, thresholds):
    """metrics.py:326-337."""
    precs = {}
    for thr in thresholds:
        per = [np.mean(e < thr) if len(e) > 0 else 0 for e in errors]
        precs[f'prec@{thr:.0e}'] = np.mean(per) if len(per) > 0 else 0
    return precs


def aggregate_metrics(metrics, epi_err_thr=5e-4):
    """metrics.py:339-377."""
    first = {}
    for idx, iden in enumerate(metrics['identifiers']):
        first[iden] = idx                                            # OrderedDict((iden, id)): the LAST index of a duplicate wins
    unq = list(first.values())
    R, t, ta = (np.asarray(metrics[k], np.float64) for k in ('R_errs', 't_errs', 't_errs_abs'))
    res = {
        'tr rot mean err': np.round(np.mean(t), 2), 'tr rot median err': np.round(np.median(t), 2),
        'tr rot pct < 30': np.round(100 * np.mean(t < 30), 1),
        'tr abs mean err': np.round(np.mean(ta), 2), 'tr abs median err': np.round(np.median(ta), 2),
        'tr abs pct < 1': np.round(100 * np.mean(ta < 1), 1),
        'rot mean err': np.round(np.mean(R), 2), 'rot median err': np.round(np.median(R), 2),
        'rot pct < 30': np.round(100 * np.mean(R < 30), 1),
        'pct successful fits': np.round(100 * np.mean(np.asarray(metrics['successful_fits'], np.float64)), 1),
        'dset size': len(t),
    }
    res.update(error_auc(np.maximum(R, t)[unq]))
    res.update(epidist_prec([metrics['epi_errs'][u] for u in unq], [epi_err_thr]))
    return res
