"""Oracle for K1: coarse matching (test infrastructure only, see oracle/__init__.py).

Follows mp3d_loftr/src/loftr/utils/coarse_matching.py: CoarseMatching.forward :86-147 (dual_softmax
branch) and get_coarse_match :149-265 (eval path), mask_border :8-25.
"""
import numpy as np

INF = 1e9


def _softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def conf_matrix(feat_c0, feat_c1, temperature, mask_c0=None, mask_c1=None, dtype=np.float32):
    """coarse_matching.py:101-118.  feat (N,L,C)/(N,S,C) float32 -> conf (N,L,S).

    dtype=float32 follows the reference's arithmetic type.  NOTE (measured, tests/test_oracle_golden.py::test_g1_coarse_full_grid_and_fp32_swamping):
    an fp32 softmax over 4800 entries where one term is ~1 and the rest ~1e-7 loses the small terms to
    swamping; the reference's own torch-CPU result (and this function in float32) deviates up to ~7e-5
    from the exact value.  dtype=float64 evaluates the same formulas on the same fp32 inputs without that
    loss and is what kernels are held to tightly; both are compared in the parity tests.
    """
    feat_c0 = np.asarray(feat_c0, np.float32)
    feat_c1 = np.asarray(feat_c1, np.float32)
    C = feat_c0.shape[-1]
    f0 = (feat_c0 / np.float32(C ** .5)).astype(dtype)                   # :104-105
    f1 = (feat_c1 / np.float32(C ** .5)).astype(dtype)
    sim = np.matmul(f0, np.swapaxes(f1, 1, 2)) / dtype(np.float32(temperature))  # :108-109 (einsum nlc,nsc->nls)
    if mask_c0 is not None:                                               # :110-113
        valid = mask_c0[..., None].astype(bool) & mask_c1[:, None].astype(bool)
        sim = np.where(valid, sim, dtype(-INF))
    return (_softmax(sim, 1) * _softmax(sim, 2)).astype(dtype)           # :114


def mask_border(m, b, v):
    """coarse_matching.py:8-25, m: (N,H0,W0,H1,W1) bool, in place."""
    if b <= 0:
        return
    m[:, :b] = v
    m[:, :, :b] = v
    m[:, :, :, :b] = v
    m[:, :, :, :, :b] = v
    m[:, -b:] = v
    m[:, :, -b:] = v
    m[:, :, :, -b:] = v
    m[:, :, :, :, -b:] = v


def get_coarse_match(conf, thr, border_rm, hw0_c, hw1_c, hw0_i, scale0=None, scale1=None):
    """coarse_matching.py:149-265, eval path (self.training False, no padded masks)."""
    N, L, S = conf.shape
    h0, w0 = hw0_c
    h1, w1 = hw1_c
    mask = conf > np.float32(thr)                                         # :174
    mask = mask.reshape(N, h0, w0, h1, w1).copy()                         # :175-176
    mask_border(mask, border_rm, False)                                   # :178
    mask = mask.reshape(N, L, S)                                          # :182-183
    mask = mask & (conf == conf.max(axis=2, keepdims=True)) \
                & (conf == conf.max(axis=1, keepdims=True))               # :186-188
    mask_v = mask.max(axis=2)                                             # :192  (first True per row)
    all_j = mask.argmax(axis=2)
    b_ids, i_ids = np.nonzero(mask_v)                                     # :193
    j_ids = all_j[b_ids, i_ids]                                           # :194
    mconf = conf[b_ids, i_ids, j_ids]                                     # :195
    scale = hw0_i[0] / hw0_c[0]                                           # :246
    s0 = scale * scale0[b_ids] if scale0 is not None else scale
    s1 = scale * scale1[b_ids] if scale1 is not None else scale
    mk0 = (np.stack([i_ids % w0, i_ids // w0], 1) * s0).astype(np.float32)  # :249-254
    mk1 = (np.stack([j_ids % w1, j_ids // w1], 1) * s1).astype(np.float32)
    keep = mconf != 0                                                     # :257-263
    return {
        'b_ids': b_ids.astype(np.int64), 'i_ids': i_ids.astype(np.int64), 'j_ids': j_ids.astype(np.int64),
        'gt_mask': mconf == 0, 'm_bids': b_ids[keep].astype(np.int64),
        'mkpts0_c': mk0[keep], 'mkpts1_c': mk1[keep], 'mconf': mconf[keep].astype(np.float32),
    }


def coarse_matching(feat_c0, feat_c1, cfg, hw0_c, hw1_c, hw0_i, mask_c0=None, mask_c1=None,
                    scale0=None, scale1=None, dtype=np.float32):
    conf = conf_matrix(feat_c0, feat_c1, cfg['dsmax_temperature'], mask_c0, mask_c1, dtype)
    out = get_coarse_match(conf, cfg['thr'], cfg['border_rm'], hw0_c, hw1_c, hw0_i, scale0, scale1)
    out['conf_matrix'] = conf
    return out


def margins(conf, thr):
    """Distances to the three discontinuities of get_coarse_match, used to filter goldens to a margin:
    |conf - thr| of the row maxima, and the relative top-2 gap of every row and column."""
    rs = np.sort(conf, axis=2)
    cs = np.sort(conf, axis=1)
    row_gap = rs[..., -1] - rs[..., -2]
    col_gap = cs[:, -1, :] - cs[:, -2, :]
    thr_gap = np.abs(rs[..., -1] - np.float32(thr))
    return row_gap, col_gap, thr_gap
