"""Coarse-level loss of the training path on the sparse confidences of K1's training kernels.

Mirrors LoFTRLoss.compute_coarse_loss (mp3d_loftr/src/losses/loftr_loss.py:56-130) for the FAR training configuration --
match_type 'dual_softmax', sparse supervision, coarse_type 'focal' -- where the reference evaluates

    conf = clamp(conf_matrix, 1e-6, 1 - 1e-6);  pos_conf = conf[conf_matrix_gt == 1]          (:84, :86-91)
    loss = c_pos_w * mean(-alpha * (1 - pos_conf)^gamma * log(pos_conf))                       (:92, :111-112)

i.e. it reads the 92 MB conf_matrix at the ground-truth positions only.  Here the positions' confidences come from
ops.coarse_pos_conf (data['conf_pos'], differentiable with a HIP backward), so neither conf_matrix nor conf_matrix_gt is
built; with a dense data['conf_matrix'] (CPU / drop-in use) the same formula indexes it.  The other loss terms of
LoFTRLoss (fine L2-with-std, 6D pose: loftr_loss.py:132-183, 247-276) are small torch expressions on per-match tensors,
restated below so that LoFTRLoss.forward (:294-356) has a counterpart with the same data-dict contract.  Pinned by
golden G15 (tools/make_goldens.py runs the reference's LoFTRLoss on the same tensors).
"""
import torch

from .pose6d import compute_normalized_6d

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


def has_no_ground_truth(data):
    """The corner case of loftr_loss.py:65-70: not a single ground-truth coarse match.  spvs_coarse then leaves ONE dummy
    entry (0, 0, 0) in spv_*_ids (supervision.py:122-128) that only keeps the fine level alive; the coarse loss must
    weigh it with zero.  far_amd's spvs_coarse records the real count; with the reference's own supervision the dense
    conf_matrix_gt tells; failing both, a lone entry at cell 0 is the dummy (cell 0 is never a real match, :103)."""
    if 'spv_gt_count' in data:
        return int(data['spv_gt_count']) == 0
    if data.get('conf_matrix_gt') is not None:
        return not bool((data['conf_matrix_gt'] == 1).any())
    ids = data['spv_i_ids']
    return ids.numel() == 0 or (ids.numel() == 1 and int(ids[0]) == 0)


def coarse_focal_loss(data, alpha=FOCAL_ALPHA, gamma=FOCAL_GAMMA, pos_weight=POS_WEIGHT, weight=None):
    """loftr_loss.py:56-112 (sparse_spvs, dual_softmax, focal).  weight: optional per-position loss weights
    (compute_c_weight, :276-283: padded-mask datasets only)."""
    p = coarse_positive_conf(data)
    if p.numel() == 0 or has_no_ground_truth(data):   # :65-70: a dummy positive with c_pos_w = 0 -> the term and its gradient vanish
        return p.sum() * 0.0 if p.numel() else (data['conf_pos'] if data.get('conf_pos') is not None else data['conf_matrix']).sum() * 0.0
    p = torch.clamp(p, 1e-6, 1 - 1e-6)                                         # :84
    loss_pos = -alpha * torch.pow(1 - p, gamma) * p.log()                      # :92
    if weight is not None:
        loss_pos = loss_pos * weight                                           # :106
    return pos_weight * loss_pos.mean()                                        # :111-112


def fine_loss_l2_std(expec_f, expec_f_gt, correct_thr=1.0, training=True):
    """loftr_loss.py:151-183 (fine_type 'l2_with_std').  expec_f (M, 3) <x, y, std>, expec_f_gt (M, 2).  Returns None in
    eval mode when no coarse match is correct (:171-172)."""
    correct = torch.linalg.norm(expec_f_gt, ord=float('inf'), dim=1) < correct_thr             # :159
    inverse_std = 1. / torch.clamp(expec_f[:, 2], min=1e-10)                                   # :162-163
    weight = (inverse_std / torch.mean(inverse_std)).detach()                                  # :164
    if not correct.any():                                                                      # :167-174
        if not training:
            return None
        correct = correct.clone()
        correct[0] = True
        weight[0] = 0.
    offset_l2 = ((expec_f_gt[correct] - expec_f[correct, :2]) ** 2).sum(-1)                    # :177
    return (offset_l2 * weight[correct]).mean()                                                # :178


def fine_loss_l2(expec_f, expec_f_gt, correct_thr=1.0, training=True):
    """loftr_loss.py:132-149 (fine_type 'l2')."""
    correct = torch.linalg.norm(expec_f_gt, ord=float('inf'), dim=1) < correct_thr
    if correct.sum() == 0:
        if not training:
            return None
        correct = correct.clone()
        correct[0] = True
    return ((expec_f_gt[correct] - expec_f[correct, :2]) ** 2).sum(-1).mean()


def rt_loss(expec_rt, T_0to1, regress_rt=True, l1=True):
    """loftr_loss.py:247-276: translation / rotation terms on the normalised 6D pose vector.  expec_rt: the head's (9,)
    output (regress_rt) or a (3, 4) pose; T_0to1 (B, 4, 4) or (B, 3, 4): only pair 0 is read (:256-261)."""
    gt = compute_normalized_6d(T_0to1[0, :3].to(expec_rt.dtype))
    pred = expec_rt if regress_rt else compute_normalized_6d(expec_rt)
    power = 1 if l1 else 2                                                                     # :264-267
    loss_tr = torch.pow(torch.abs(pred[:3] - gt[:3]), power).mean()                            # :270
    loss_rot = torch.pow(torch.abs(pred[3:] - gt[3:]), power).mean()                           # :272
    return torch.clamp(loss_tr, 1e-8, 1e5), torch.clamp(loss_rot, 1e-8, 1e5)                   # :274-275


class LoFTRLoss(torch.nn.Module):
    """Counterpart of src/losses/loftr_loss.py:LoFTRLoss for the FAR training configurations (dual_softmax, focal coarse
    loss, sparse supervision; l2_with_std fine loss; 6D pose loss).  Same constructor argument (the lower-cased config
    with ['loftr']['loss'], ['loftr']['match_coarse'], ...) and the same effect: forward(data) writes data['loss'] and
    data['loss_scalars'].  On the GPU training path the coarse term reads data['conf_pos'] (K1's sparse HIP kernels);
    with a dense data['conf_matrix'] it gathers the same positions from it (given as spv ids or as conf_matrix_gt)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        lc = self.loss_config = config['loftr']['loss']
        mc = config['loftr']['match_coarse']
        if mc['match_type'] != 'dual_softmax' or lc['coarse_type'] != 'focal' or not mc.get('sparse_spvs', True):
            raise NotImplementedError('far_amd.losses.LoFTRLoss covers dual_softmax + focal + sparse supervision '
                                      '(the FAR training scripts); sinkhorn / cross-entropy / dense supervision are not built')
        self.correct_thr = lc['fine_correct_thr']
        self.c_pos_w = lc['pos_weight']
        self.fine_type = lc['fine_type']

    @torch.no_grad()
    def compute_c_weight(self, data):
        """:184-191, reduced to the ground-truth positions the sparse loss reads (weight[pos_mask])."""
        if 'mask0' not in data:
            return None
        m0 = data['mask0'].flatten(-2).float()
        m1 = data['mask1'].flatten(-2).float()
        return m0[data['spv_b_ids'], data['spv_i_ids']] * m1[data['spv_b_ids'], data['spv_j_ids']]

    def forward(self, data):
        cfg = self.config
        lc = self.loss_config
        ref = data['conf_pos'] if data.get('conf_pos') is not None else data.get('conf_matrix')
        dev = ref.device if ref is not None else data['expec_rt'].device
        loss = torch.zeros(1, device=dev)                                                      # :303
        scalars = {}
        if cfg['loftr'].get('from_saved_preds') is None and not cfg.get('use_correspondence_transformer', False):
            d = data
            if data.get('conf_pos') is None and 'spv_b_ids' not in data:                       # dense drop-in use: positions from conf_matrix_gt
                b, i, j = torch.where(data['conf_matrix_gt'] == 1)
                d = dict(data, spv_b_ids=b, spv_i_ids=i, spv_j_ids=j, spv_gt_count=int(b.numel()))
            loss_c = coarse_focal_loss(d, lc['focal_alpha'], lc['focal_gamma'], self.c_pos_w, weight=self.compute_c_weight(d))
            loss = loss + loss_c * lc['coarse_weight']                                         # :314
            scalars['loss_c'] = loss_c.detach().cpu()
            fn = fine_loss_l2_std if self.fine_type == 'l2_with_std' else fine_loss_l2
            loss_f = fn(data['expec_f'], data['expec_f_gt'], self.correct_thr, self.training)  # :318
            if loss_f is not None:
                loss = loss + loss_f * lc['fine_weight']
                scalars['loss_f'] = loss_f.detach().cpu()
            else:
                assert self.training is False
                scalars['loss_f'] = torch.tensor(1.)
        if (lc['rt_weight_tr'] + lc['rt_weight_rot']) > 0 and data.get('expec_rt') is not None:   # :327
            l_tr, l_rot = rt_loss(data['expec_rt'], data['T_0to1'][:, :3], cfg['loftr']['regress_rt'], lc.get('use_l1_rt_loss', False))
            loss = loss + l_tr * lc['rt_weight_tr'] + l_rot * lc['rt_weight_rot']
            scalars.update(loss_rot=l_rot.detach().cpu(), loss_tr=l_tr.detach().cpu())
        else:
            scalars.update(loss_rot=torch.tensor(100.), loss_tr=torch.tensor(4.))
        if cfg['loftr'].get('predict_translation_scale', False):
            raise NotImplementedError('predict_translation_scale is off in every FAR script (DESIGN.md section 9)')
        for k in ('num_correspondences_after_ransac', 'num_correspondences_before_ransac'):   # :340-351
            v = data.get(k, 0)
            scalars[k] = v.detach().cpu() if torch.is_tensor(v) else torch.tensor(v, dtype=torch.float32)
        scalars['loss'] = loss.detach().cpu()
        data.update(loss=loss, loss_scalars=scalars)
