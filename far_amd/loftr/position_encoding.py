"""2-D sinusoidal position encoding added to the 1/8 feature map.
Mirrors mp3d_loftr/src/loftr/utils/position_encoding.py:6-42 (same buffer name `pe`, non-persistent)."""
import math

import torch
from torch import nn


class PositionEncodingSine(nn.Module):
    def __init__(self, d_model, max_shape=(256, 256), temp_bug_fix=True):
        super().__init__()
        ys = torch.ones(max_shape).cumsum(0).float().unsqueeze(0)
        xs = torch.ones(max_shape).cumsum(1).float().unsqueeze(0)
        k = torch.arange(0, d_model // 2, 2).float()
        if temp_bug_fix:
            div = torch.exp(k * (-math.log(10000.0) / (d_model // 2)))
        else:  # the historical operator-precedence variant kept by the reference (:28-29)
            div = torch.exp(k * (-math.log(10000.0) / d_model // 2))
        div = div[:, None, None]
        pe = torch.zeros((d_model, *max_shape))
        pe[0::4] = torch.sin(xs * div)
        pe[1::4] = torch.cos(xs * div)
        pe[2::4] = torch.sin(ys * div)
        pe[3::4] = torch.cos(ys * div)
        self.register_buffer('pe', pe.unsqueeze(0), persistent=False)

    def forward(self, x):
        return x + self.pe[:, :, :x.size(2), :x.size(3)]
