"""Coarse-level loss of the training path on the sparse confidences of K1's training kernels.

Mirrors LoFTRLoss.compute_coarse_loss (mp3d_loftr/src/losses/loftr_loss.py:56-130) for the FAR training configuration --
match_type 'dual_softmax', sparse supervision, coarse_type 'focal' -- where the reference evaluates

    conf = clamp(conf_matrix, 1e-6, 1 - 1e-6);  pos_conf = conf[conf_matrix_gt == 1]          (:84, :86-91)
    loss = c_pos_w * mean(-alpha * (1 - pos_conf)^gamma * log(pos_conf))                       (:92, :111-112)

i.e. it reads the 92 MB conf_matrix at the ground-truth positions only.  Here the positions' confidences come from
ops.coarse_pos_conf (data['conf_pos'], differentiable with a HIP backward), so neither conf_matrix nor conf_matrix_gt is
built; with a dense data['conf_matrix'] (CPU / drop-in use) the same formula indexes it.  The other loss terms of
LoFTRLoss (fine L2-with-std, 6D pose) are small torch expressions on per-match tensors and stay as the reference has
them (out of scope).
"""
import torch

# loss defaults of the reference configuration (src/config/default.py: LOFTR.LOSS.*)
FOCAL_ALPHA = 0.25
FOCAL_GAMMA = 2.0
POS_WEIGHT = 1.0


def coarse_positive_conf(data):
    """(M,) confidences at the ground-truth coarse matches: data['conf_pos'] (GPU training path) or the gather from a
    dense conf_matrix (conf[pos_mask] visits the positives in (b, i, j) order; the mean below is order-independent)."""
    if data.get('conf_pos') is not None:
        return data['conf_pos']
    return data['conf_matrix'][data['spv_b_ids'], data['spv_i_ids'], data['spv_j_ids']]


def coarse_focal_loss(data, alpha=FOCAL_ALPHA, gamma=FOCAL_GAMMA, pos_weight=POS_WEIGHT, weight=None):
    """loftr_loss.py:56-112 (sparse_spvs, dual_softmax, focal).  weight: optional per-position loss weights
    (compute_c_weight, :276-283: padded-mask datasets only)."""
    p = coarse_positive_conf(data)
    if p.numel() == 0:                       # corner case :64-68: no ground-truth match -> a dummy positive with weight 0
        ref = data['conf_pos'] if data.get('conf_pos') is not None else data['conf_matrix']
        return ref.sum() * 0.0
    p = torch.clamp(p, 1e-6, 1 - 1e-6)                                         # :84
    loss_pos = -alpha * torch.pow(1 - p, gamma) * p.log()                      # :92
    if weight is not None:
        loss_pos = loss_pos * weight                                           # :106
    return pos_weight * loss_pos.mean()                                        # :111-112
