"""ResNet-18-style FPN producing the 1/8 (256 ch) and 1/2 (128 ch) maps.

Mirrors the architecture and PARAMETER NAMES of mp3d_loftr/src/loftr/backbone/resnet_fpn.py:15-119
(BasicBlock, ResNetFPN_8_2) so reference checkpoints load.  The convolutions stay on the vendor path
(MIOpen through torch) -- SURVEY.md section 2.1 #2: not a custom kernel; run it channels_last / bf16 for speed.
"""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _fold(bn):
    """Inference BatchNorm as a per-channel affine map: y = x * scale + shift."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()


def _fused_ok(m, x):
    """The fused epilogues (K7/K8) apply to inference on fp32 channels_last GPU tensors only."""
    return (not m.training) and (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32


def _c1(i, o, s=1):
    return nn.Conv2d(i, o, 1, stride=s, padding=0, bias=False)


def _c3(i, o, s=1):
    return nn.Conv2d(i, o, 3, stride=s, padding=1, bias=False)


class BasicBlock(nn.Module):
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = _c3(in_planes, planes, stride)
        self.conv2 = _c3(planes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None if stride == 1 else nn.Sequential(_c1(in_planes, planes, stride), nn.BatchNorm2d(planes))

    def forward(self, x):
        if _fused_ok(self, x):
            y = ops.affine_act(self.conv1(x), *_fold(self.bn1), act='relu')
            y = self.conv2(y)
            if self.downsample is not None:
                x = ops.affine_act(self.downsample[0](x), *_fold(self.downsample[1]), act='none')
            return ops.affine_act(y, *_fold(self.bn2), residual=x, act='relu')       # relu(x + bn2(conv2(.)))
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class ResNetFPN_8_2(nn.Module):
    def __init__(self, config):
        super().__init__()
        d0 = config['initial_dim']
        b = config['block_dims']
        self.config = config
        self.conv1 = nn.Conv2d(1, d0, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(d0)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(BasicBlock(d0, b[0], 1), BasicBlock(b[0], b[0], 1))      # 1/2
        self.layer2 = nn.Sequential(BasicBlock(b[0], b[1], 2), BasicBlock(b[1], b[1], 1))    # 1/4
        self.layer3 = nn.Sequential(BasicBlock(b[1], b[2], 2), BasicBlock(b[2], b[2], 1))    # 1/8
        self.layer3_outconv = _c1(b[2], b[2])
        self.layer2_outconv = _c1(b[1], b[2])
        self.layer2_outconv2 = nn.Sequential(_c3(b[2], b[2]), nn.BatchNorm2d(b[2]), nn.LeakyReLU(), _c3(b[2], b[1]))
        self.layer1_outconv = _c1(b[0], b[1])
        self.layer1_outconv2 = nn.Sequential(_c3(b[1], b[1]), nn.BatchNorm2d(b[1]), nn.LeakyReLU(), _c3(b[1], b[0]))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    # Optional reduced precision for the fine (1/2-resolution) FPN branch only.  The coarse map x3_out -- the only
    # input of the discrete matching decisions -- always comes from the fp32 trunk; the fine map only feeds the
    # sub-pixel expectation.  None (default) = everything fp32 = the parity configuration.
    fine_branch_dtype = None

    def _trunk(self, x):
        if _fused_ok(self, x):
            x0 = ops.affine_act(self.conv1(x), *_fold(self.bn1), act='relu')
        else:
            x0 = self.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x0)            # BasicBlocks pick their own fused / plain path
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        return x1, x2, x3

    def _fpn_fused(self, x1, x2, x3_out):
        o2, o1 = self.layer2_outconv2, self.layer1_outconv2
        y = ops.upsample2x_add(x3_out, self.layer2_outconv(x2))
        y = ops.affine_act(o2[0](y), *_fold(o2[1]), act='leaky', slope=o2[2].negative_slope)
        x2_out = o2[3](y)
        y = ops.upsample2x_add(x2_out, self.layer1_outconv(x1))
        y = ops.affine_act(o1[0](y), *_fold(o1[1]), act='leaky', slope=o1[2].negative_slope)
        return o1[3](y)

    def _fpn_plain(self, x1, x2, x3_out):
        up3 = F.interpolate(x3_out, scale_factor=2., mode='bilinear', align_corners=True)
        x2_out = self.layer2_outconv2(self.layer2_outconv(x2) + up3)
        up2 = F.interpolate(x2_out, scale_factor=2., mode='bilinear', align_corners=True)
        return self.layer1_outconv2(self.layer1_outconv(x1) + up2)

    def forward(self, x):
        x1, x2, x3 = self._trunk(x)
        x3_out = self.layer3_outconv(x3)
        dt = self.fine_branch_dtype
        if dt is None and _fused_ok(self, x3_out):
            return [x3_out, self._fpn_fused(x1, x2, x3_out)]
        with (torch.autocast('cuda', dtype=dt) if dt is not None else contextlib.nullcontext()):
            return [x3_out, self._fpn_plain(x1, x2, x3_out)]


def build_backbone(config):
    if config['backbone_type'] == 'ResNetFPN' and tuple(config['resolution']) == (8, 2):
        return ResNetFPN_8_2(config['resnetfpn'])
    raise ValueError(f"backbone {config['backbone_type']} / resolution {config['resolution']} not supported")
