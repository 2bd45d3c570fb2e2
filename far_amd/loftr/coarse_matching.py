"""CoarseMatching on kernel K1.  Mirrors the interface of mp3d_loftr/src/loftr/utils/coarse_matching.py
(CoarseMatching :56-265): same constructor dict, same data-dict keys written."""
import torch
import torch.nn as nn

from .. import ops


class CoarseMatching(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.thr = config['thr']
        self.border_rm = config['border_rm']
        self.train_coarse_percent = config['train_coarse_percent']
        self.train_pad_num_gt_min = config['train_pad_num_gt_min']
        self.match_type = config['match_type']
        if self.match_type != 'dual_softmax':
            raise NotImplementedError("only match_type='dual_softmax' (the FAR configuration) has a kernel")
        self.temperature = config['dsmax_temperature']
        # data['conf_matrix'] (92 MB / pair) is consumed only by the coarse loss and by plotting
        # (loftr_loss.py:307-311); it is materialised when training or when asked for explicitly.
        self.materialize_conf = False

    def forward(self, feat_c0, feat_c1, data, mask_c0=None, mask_c1=None):
        """feat_c0 [N, L, C], feat_c1 [N, S, C]; updates data with conf_matrix (optional), b_ids, i_ids,
        j_ids, gt_mask, m_bids, mkpts0_c, mkpts1_c, mconf (coarse_matching.py:144-147, :243-263)."""
        if self.training:
            raise NotImplementedError('training-time sampling/padding (coarse_matching.py:199-240) and the '
                                      'backward of K1 are not implemented in this round')
        hw0, hw1 = data['hw0_c'], data['hw1_c']
        valid_hw = None
        as_u8 = lambda m: None if m is None else m.to(torch.uint8).contiguous()
        if 'mask0' in data:
            # mask_border_with_padding (:28-43): per-sample valid extents of the padded coarse masks
            m0, m1 = data['mask0'], data['mask1']
            valid_hw = torch.stack([m0.sum(1).max(-1)[0], m0.sum(-1).max(-1)[0],
                                    m1.sum(1).max(-1)[0], m1.sum(-1).max(-1)[0]], 1).to(torch.int32).contiguous()
        scale = data['hw0_i'][0] / data['hw0_c'][0]
        s0 = data['scale0'].float().contiguous() if 'scale0' in data else None
        s1 = data['scale1'].float().contiguous() if 'scale1' in data else None
        out = ops.coarse_match(feat_c0.float().contiguous(), feat_c1.float().contiguous(), self.temperature,
                               self.thr, self.border_rm, hw0, hw1, scale, as_u8(mask_c0), as_u8(mask_c1),
                               valid_hw, s0, s1, want_conf=self.materialize_conf)
        mconf = out['mconf']
        data.update({
            'conf_matrix': out['conf_matrix'],
            'b_ids': out['b_ids'], 'i_ids': out['i_ids'], 'j_ids': out['j_ids'],
            'gt_mask': mconf == 0,
            'm_bids': out['b_ids'],          # eval: every match has mconf > thr > 0, nothing is dropped (:257-263)
            'mkpts0_c': out['mkpts0_c'], 'mkpts1_c': out['mkpts1_c'], 'mconf': mconf,
            'match_counts': out['counts'],   # per-pair M (host), reused by the batched solver
        })
