"""Oracle for K5 and the LoFTR encoder layer (test infrastructure only).

Follows mp3d_loftr/src/loftr/loftr_module/linear_attention.py:20-52 and
mp3d_loftr/src/loftr/loftr_module/transformer.py:44-67 (LoFTREncoderLayer.forward), :90-112.
"""
import numpy as np


def elu1(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))) + 1      # elu(x) + 1 (linear_attention.py:9-10)


def linear_attention(q, k, v, nhead, q_mask=None, kv_mask=None, eps=1e-6, dtype=np.float32):
    """q (N,L,C), k, v (N,S,C) raw projections -> (N,L,C).  linear_attention.py:31-50."""
    N, L, C = q.shape
    S = k.shape[1]
    D = C // nhead
    Q = elu1(q.astype(dtype)).reshape(N, L, nhead, D)
    K = elu1(k.astype(dtype)).reshape(N, S, nhead, D)
    V = v.astype(dtype).reshape(N, S, nhead, D)
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    if kv_mask is not None:
        K = K * kv_mask[:, :, None, None]
        V = V * kv_mask[:, :, None, None]
    V = V / dtype(S)                                                # :43
    KV = np.einsum('nshd,nshv->nhdv', K, V)                         # :44
    Z = 1 / (np.einsum('nlhd,nhd->nlh', Q, K.sum(1)) + dtype(eps))  # :45
    out = np.einsum('nlhd,nhdv,nlh->nlhv', Q, KV, Z) * dtype(S)     # :50
    return out.reshape(N, L, C).astype(dtype)
