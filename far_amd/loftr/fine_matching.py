"""FineMatching on kernel K3b.  Mirrors mp3d_loftr/src/loftr/utils/fine_matching.py:8-76."""
import math

import torch
import torch.nn as nn

from .. import autograd_ops as ag
from .. import ops


class FineMatching(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config

    def forward(self, feat_f0, feat_f1, data, train=False):
        M, WW, C = feat_f0.shape
        W = int(math.sqrt(WW))
        scale = data['hw0_i'][0] / data['hw0_f'][0]
        if M == 0:                                                                   # :33-41
            assert self.training is False, "M is always >0 when training (coarse_matching.py)"
            data.update({'expec_f': torch.empty(0, 3, device=feat_f0.device),
                         'mkpts0_f': data['mkpts0_c'], 'mkpts1_f': data['mkpts1_c']})
            return
        if ag.needs_grad(feat_f0, feat_f1):
            coords, std = ag.fine_expect(feat_f0, feat_f1)
            data.update({'expec_f': torch.cat([coords, std.unsqueeze(1)], -1)})
            if not self.config['regress_rt'] or not train or self.config['regress']['use_simple_moe']:
                with torch.no_grad():
                    sc1 = scale * data['scale1'][data['b_ids']] if 'scale0' in data else scale
                    n = len(data['mconf'])
                    data.update({'mkpts0_f': data['mkpts0_c'],
                                 'mkpts1_f': data['mkpts1_c'] + (coords * (W // 2) * sc1)[:n]})
            return
        s1 = data['scale1'].float().contiguous() if 'scale0' in data else None       # :70
        expec, mk1 = ops.fine_expect(feat_f0.float().contiguous(), feat_f1.float().contiguous(),
                                     data['mkpts1_c'].contiguous(), (W // 2) * scale, s1,
                                     data['b_ids'] if s1 is not None else None)
        data.update({'expec_f': expec})
        if not self.config['regress_rt'] or not train or self.config['regress']['use_simple_moe']:   # :59-62
            n = len(data['mconf'])
            data.update({'mkpts0_f': data['mkpts0_c'], 'mkpts1_f': mk1[:n]})
