"""FinePreprocess on kernel K3a.  Mirrors mp3d_loftr/src/loftr/loftr_module/fine_preprocess.py:7-59
(same parameters: down_proj, merge_feat)."""
import torch
import torch.nn as nn

from .. import autograd_ops as ag
from .. import ops


class FinePreprocess(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.cat_c_feat = config['fine_concat_coarse_feat']
        self.W = config['fine_window_size']
        d_c, d_f = config['coarse']['d_model'], config['fine']['d_model']
        self.d_model_f = d_f
        if self.cat_c_feat:
            self.down_proj = nn.Linear(d_c, d_f, bias=True)
            self.merge_feat = nn.Linear(2 * d_f, d_f, bias=True)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")

    def forward(self, feat_f0, feat_f1, feat_c0, feat_c1, data):
        W = self.W
        stride = data['hw0_f'][0] // data['hw0_c'][0]
        data.update({'W': W})
        b, i, j = data['b_ids'], data['i_ids'], data['j_ids']
        if b.shape[0] == 0:                                                          # :34-37
            e = torch.empty(0, W ** 2, self.d_model_f, device=feat_f0.device)
            return e, e.clone()
        # direct gather of the M x 25 x C window values instead of unfolding both full maps (:40-47)
        if ag.needs_grad(feat_f0, feat_f1):
            w0 = ag.fine_windows(feat_f0, b, i, W, stride)
            w1 = ag.fine_windows(feat_f1, b, j, W, stride)
        else:
            w0 = ops.fine_gather(feat_f0.float(), b, i, data['hw0_c'][1], W, stride)
            w1 = ops.fine_gather(feat_f1.float(), b, j, data['hw1_c'][1], W, stride)
        if self.cat_c_feat:
            c_win = self.down_proj(torch.cat([feat_c0[b, i], feat_c1[b, j]], 0))     # [2M, C]
            both = torch.cat([torch.cat([w0, w1], 0), c_win.unsqueeze(1).expand(-1, W ** 2, -1)], -1)
            w0, w1 = torch.chunk(self.merge_feat(both), 2, dim=0)
        return w0, w1
