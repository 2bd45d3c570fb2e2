"""Oracle for K3: fine-level window extraction and sub-pixel refinement (test infrastructure only).

Follows mp3d_loftr/src/loftr/loftr_module/fine_preprocess.py:29-59 and
mp3d_loftr/src/loftr/utils/fine_matching.py:15-76.  kornia's create_meshgrid / spatial_expectation2d
(kornia 0.7.1, not vendored, not installed) are restated from their published definitions: parity unpinned
against kornia itself; anchored on the reference call sites fine_matching.py:49-50.
"""
import numpy as np


def unfold_windows(feat_f, b_ids, cell_ids, wc, W, stride):
    """fine_preprocess.py:40-47.  feat_f (N,C,Hf,Wf) -> (M, W*W, C); F.unfold ordering '(c ww) l'."""
    N, C, Hf, Wf = feat_f.shape
    pad = W // 2
    fp = np.zeros((N, C, Hf + 2 * pad, Wf + 2 * pad), feat_f.dtype)
    fp[:, :, pad:pad + Hf, pad:pad + Wf] = feat_f
    M = len(b_ids)
    out = np.zeros((M, W * W, C), feat_f.dtype)
    for m in range(M):
        y0 = (int(cell_ids[m]) // wc) * stride
        x0 = (int(cell_ids[m]) % wc) * stride
        win = fp[int(b_ids[m]), :, y0:y0 + W, x0:x0 + W]          # (C, W, W)
        out[m] = win.reshape(C, W * W).T
    return out


def meshgrid_normalized(W):
    """kornia.utils.create_meshgrid(W, W, True) reshaped to (W*W, 2): [..., 0] = x (fastest), [..., 1] = y."""
    xs = np.linspace(-1, 1, W, dtype=np.float32)
    gy, gx = np.meshgrid(xs, xs, indexing='ij')
    return np.stack([gx.reshape(-1), gy.reshape(-1)], 1).astype(np.float32)


def fine_matching(feat_f0, feat_f1, mkpts1_c, W_scale, dtype=np.float32):
    """fine_matching.py:43-54 + :64-76.  feat (M,WW,C).  Returns expec_f (M,3), mkpts1_f (M,2)."""
    M, WW, C = feat_f0.shape
    W = int(np.sqrt(WW))
    f0 = feat_f0.astype(dtype)
    f1 = feat_f1.astype(dtype)
    picked = f0[:, WW // 2, :]                                     # :43
    sim = np.einsum('mc,mrc->mr', picked, f1)                      # :44
    temp = dtype(1. / C ** .5)                                     # :45
    x = temp * sim
    x = x - x.max(1, keepdims=True)
    heat = np.exp(x)
    heat = heat / heat.sum(1, keepdims=True)                       # :46
    grid = meshgrid_normalized(W).astype(dtype)                    # :50
    coords = heat @ grid                                           # :49 spatial_expectation2d -> (x, y)
    var = heat @ (grid ** 2) - coords ** 2                         # :53
    std = np.sqrt(np.clip(var, 1e-10, None)).sum(-1)               # :54
    expec = np.concatenate([coords, std[:, None]], -1)             # :57
    mk1 = np.asarray(mkpts1_c, dtype) + coords * dtype(W_scale)    # :71
    return expec.astype(dtype), mk1.astype(dtype)
