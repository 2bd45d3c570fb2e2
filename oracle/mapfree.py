"""Oracle for K12: the correlation-volume warp of the Map-free 6DReg aggregator (test infrastructure only, see
oracle/__init__.py).

Follows mapfree_6dreg/lib/models/regression/aggregator.py:44-115 (CorrelationVolumeWarping.forward) in the FAR
configuration (config/regression/mapfree/rot6d_trans_with_loftr.yaml): POSITION_ENCODER and MAX_SCORE_CHANNEL on; no
dustbin, no feature normalisation, no half channels, no extra blocks.  Pinned by golden G13 (tools/make_goldens.py runs
the reference module itself).
"""
import numpy as np


def meshgrid_uv(H, W, dtype=np.float32):
    """aggregator.py:81-84: u = linspace(-1, 1, H) along rows, v = linspace(-1, 1, W) along columns, as (2, H*W)."""
    u = np.linspace(-1, 1, H, dtype=np.float32).astype(dtype)
    v = np.linspace(-1, 1, W, dtype=np.float32).astype(dtype)
    uu, vv = np.meshgrid(u, v, indexing='ij')
    return np.stack([uu.reshape(-1), vv.reshape(-1)], 0)


def corr_volume_warp(vol0, vol1, dtype=np.float64):
    """vol0, vol1 (B, D, H, W) float32 -> agg (B, 2 D + 3, H, W): cat[vol0, vol1w, pos_encoder, max_score]."""
    B, D, H, W = vol0.shape
    a = vol0.reshape(B, D, H * W).astype(dtype)
    b = vol1.reshape(B, D, H * W).astype(dtype)
    cv = np.einsum('bdi,bdj->bij', a, b)                                  # :58 bmm(vol0^T, vol1)
    cv = cv - cv.max(axis=2, keepdims=True)
    cv = np.exp(cv)
    cv /= cv.sum(axis=2, keepdims=True)                                   # :69 softmax(dim=2)
    vol1w = np.einsum('bdj,bij->bdi', b, cv)                              # :72 bmm(vol1, cvolume^T)
    grid = meshgrid_uv(H, W, dtype)
    pos = np.einsum('gj,bij->bgi', grid, cv)                              # :86
    mx = cv.max(axis=2)[:, None, :]                                       # :104-105
    return np.concatenate([a, vol1w, pos, mx], 1).reshape(B, 2 * D + 3, H, W)
