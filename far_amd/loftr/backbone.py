"""ResNet-18-style FPN producing the 1/8 (256 ch) and 1/2 (128 ch) maps.

Mirrors the architecture and PARAMETER NAMES of mp3d_loftr/src/loftr/backbone/resnet_fpn.py:15-119
(BasicBlock, ResNetFPN_8_2) so reference checkpoints load.

Inference in fp32 on the GPU runs `_forward_fused`: NHWC activations end to end, the stem on K10, every 3x3 / 1x1
convolution on K9 (split-fp16 implicit GEMM with BatchNorm, activation and residual add fused into the epilogue), the
FPN upsample-add on K8: no vendor convolution is left on the inference path.
Training (gradients) on the GPU runs the same module graph with every 3x3 / 1x1 convolution on K9 forward and K9 dgrad
(ops.conv_train), BatchNorm with batch statistics + activation + shortcut add on K19; CPU tensors and `hip_training = False`
need the test-side helper (far_amd/_vendor.py) -- the plain torch modules are then what runs.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _vendor, flags, ops


def _fold(bn):
    """Inference BatchNorm as a per-channel affine map: y = x * scale + shift."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()


def _fused_ok(m, x):
    """The kernel path (K10 / K9 / K8 / K7) serves inference on fp32 GPU tensors."""
    return (not m.training) and (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32


class _PackCache(ops.PackCache):
    """K9 weight images per convolution with the following BatchNorm folded into the epilogue vectors."""

    pad = False          # set by ResNetFPN_8_2._forward_fused: feature maps padded to multiples of 16 channels in HBM

    def get(self, key, conv, bn=None, split=True):
        ts = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

        def build():
            scale, shift = _fold(bn) if bn is not None else (None, None)
            stride = conv.stride[0] if conv.kernel_size[0] == 3 else 1      # 1x1 stride 2: the caller subsamples
            w = conv.weight.detach()
            co, ci = w.shape[:2]
            cop, cip = (_pad16(co), _pad16(ci)) if self.pad else (co, ci)
            if (cop, cip) != (co, ci):
                # the 196-channel maps live in HBM with 208 channels (832-byte pixels, 64-byte aligned k-step chunks): zero weights
                # for the extra input channels, zero weights / unit scale / zero shift for the extra output channels, which
                # therefore hold exact zeros through ReLU, LeakyReLU, residual adds and the FPN merge.  Same matrix work (the
                # kernels pad a k-step to 16 channels anyway), bit-identical real channels; K17 196 -> 196 @240x320 8.78 ->
                # 8.21 ms (profiles/r04_pmc_traffic.json, tools/stride_ab.py)
                wp = w.new_zeros(cop, cip, *w.shape[2:])
                wp[:co, :ci] = w
                w = wp
                if scale is not None:
                    scale = torch.cat([scale.detach(), scale.new_ones(cop - co)])
                    shift = torch.cat([shift.detach(), shift.new_zeros(cop - co)])
            return ops.PackedConv(w, scale, shift, split=split, stride=stride)
        return super().get((key, split, self.pad), ts, build)


def _pad16(c):
    """Channel count of a feature map in HBM on the inference path: the next multiple of 16 (196 -> 208)."""
    return -(-c // 16) * 16


PAD_CHANNELS = not flags.off('FAR_NO_PAD')


def _c1(i, o, s=1):
    return nn.Conv2d(i, o, 1, stride=s, padding=0, bias=False)


def _c3(i, o, s=1):
    return nn.Conv2d(i, o, 3, stride=s, padding=1, bias=False)


def _bn_hip(bn, x):
    """BatchNorm with batch statistics on K19: a plain nn.BatchNorm2d in training mode (NOT SyncBatchNorm -- the reference's multi-GPU
    training converts the modules, train.py:342: their statistics span the ranks and stay with torch), fp32 GPU tensor, gradients
    enabled."""
    return (ResNetFPN_8_2.hip_training and ops.USE_HIP_BATCHNORM_TRAIN and type(bn) is nn.BatchNorm2d and bn.training and x.is_cuda
            and x.dtype == torch.float32 and torch.is_grad_enabled())


def _conv(conv, x, owner):
    """conv(x) for the reference-style (training / CPU) forward: on the GPU with gradients enabled the bias-free 3x3 / 1x1
    convolutions run K9 forward and K9 dgrad (ops.conv_train); otherwise the nn.Conv2d itself."""
    if (ResNetFPN_8_2.hip_training and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and conv.bias is None
            and conv.kernel_size[0] in (1, 3) and conv.in_channels % 4 == 0 and (x.requires_grad or conv.weight.requires_grad)):
        pk = owner.__dict__.setdefault('_train_packs', ops.PackCache())
        return ops.conv_train(x, conv.weight, conv.stride[0], pk, id(conv))
    return conv(x)


class BasicBlock(nn.Module):
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = _c3(in_planes, planes, stride)
        self.conv2 = _c3(planes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None if stride == 1 else nn.Sequential(_c1(in_planes, planes, stride), nn.BatchNorm2d(planes))

    def forward(self, x):          # the module graph of training (inference runs _forward_fused)
        if _bn_hip(self.bn1, x):   # training on the GPU: BatchNorm (batch statistics) + activation + shortcut add on K19 / K7
            h = ops.bn_act_train(_conv(self.conv1, x, self), self.bn1, 'relu')
            if self.downsample is not None:
                x = ops.bn_act_train(_conv(self.downsample[0], x, self), self.downsample[1], 'none')
            return ops.bn_act_train(_conv(self.conv2, h, self), self.bn2, 'relu', residual=x)
        y = self.bn2(_conv(self.conv2, self.relu(self.bn1(_conv(self.conv1, x, self))), self))
        if self.downsample is not None:
            x = self.downsample[1](_conv(self.downsample[0], x, self))
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

    hip_training = True          # training on the GPU: convolutions on K9 (forward + dgrad); False: vendor convolutions + autograd

    def _outconv2(self, seq, x):
        if _bn_hip(seq[1], x):
            return _conv(seq[3], ops.bn_act_train(_conv(seq[0], x, self), seq[1], 'leaky', seq[2].negative_slope), self)
        return _conv(seq[3], seq[2](seq[1](_conv(seq[0], x, self))), self)

    def _merge(self, lateral, coarse):
        """lateral + 2x bilinear upsampling of the coarser level (:108-109, :113-114)."""
        if (ResNetFPN_8_2.hip_training and lateral.is_cuda and lateral.dtype == torch.float32 and torch.is_grad_enabled()
                and lateral.shape[1] % 4 == 0 and tuple(lateral.shape[2:]) == (2 * coarse.shape[2], 2 * coarse.shape[3])):
            return ops.upsample2x_add_train(coarse, lateral)          # K8 + its fixed-order gradient
        return lateral + F.interpolate(coarse, scale_factor=2., mode='bilinear', align_corners=True)

    def _fpn_plain(self, x1, x2, x3_out):
        x2_out = self._outconv2(self.layer2_outconv2, self._merge(_conv(self.layer2_outconv, x2, self), x3_out))
        return self._outconv2(self.layer1_outconv2, self._merge(_conv(self.layer1_outconv, x1, self), x2_out))

    # ---- inference fast path: NHWC activations, K10 + K9 + K8 (tensors below are (N, H, W, C)) ----------------
    # K9 operand precision: True = split fp16 pairs (fp32-grade, the parity configuration), False = plain fp16.
    # The coarse map -- the only input of the discrete matching decisions -- depends on the trunk + layer3_outconv
    # only; the FPN branch feeds the sub-pixel refinement.
    trunk_split = True
    fpn_split = True

    def _block_fused(self, name, blk, x, pk):
        sp = self.trunk_split
        y = ops.conv_nhwc(x, pk.get(name + '.conv1', blk.conv1, blk.bn1, sp), act='relu')      # stride 1 or 2
        if blk.downsample is None:
            res = x
        else:        # 1x1 stride-2 shortcut: K9 on the subsampled pixels, its BatchNorm folded
            res = ops.conv_nhwc(x, pk.get(name + '.down', blk.downsample[0], blk.downsample[1], sp), in_stride=2)   # x[:, ::2, ::2] in place
        return ops.conv_nhwc(y, pk.get(name + '.conv2', blk.conv2, blk.bn2, sp), residual=res, act='relu')

    def _forward_fused(self, x, side=None):
        pk = self.__dict__.setdefault('_packs', _PackCache())
        # only the middle level may need padding: the stem's and the two returned maps' channel counts are what the callers see
        b = self.config['block_dims']
        pk.pad = PAD_CHANNELS and self.config['initial_dim'] % 16 == 0 and b[0] % 16 == 0 and b[2] % 16 == 0
        sp = self.trunk_split
        # (the returned maps are (N, C, H, W)-shaped views of NHWC buffers: downstream 'n c h w -> n (h w) c' is then a free view)
        x0 = ops.stem7x7(x, self.conv1.weight, *_fold(self.bn1))
        x1 = self._block_fused('layer1.1', self.layer1[1], self._block_fused('layer1.0', self.layer1[0], x0, pk), pk)
        x2 = self._block_fused('layer2.1', self.layer2[1], self._block_fused('layer2.0', self.layer2[0], x1, pk), pk)
        x3 = self._block_fused('layer3.1', self.layer3[1], self._block_fused('layer3.0', self.layer3[0], x2, pk), pk)
        x3_out = ops.conv_nhwc(x3, pk.get('layer3_outconv', self.layer3_outconv, None, sp))
        if side is None:
            return [x3_out.permute(0, 3, 1, 2), self._fpn_fused(x1, x2, x3_out, pk).permute(0, 3, 1, 2)]
        # The FPN's fine branch (21 of the backbone's 48 ms per 32 pairs) feeds the fine level only; the coarse transformer and K1
        # need x3_out alone.  With a side stream from the caller the branch forks off here and the caller joins it (the returned
        # event) in front of its first reader.  Tensors that cross streams are recorded with the allocator on the other side.
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record()
        side.wait_event(fork)
        with torch.cuda.stream(side):
            x1_out = self._fpn_fused(x1, x2, x3_out, pk)
            done = torch.cuda.Event()
            done.record()
        for t in (x1, x2, x3_out):
            t.record_stream(side)
        x1_out.record_stream(main)
        return [x3_out.permute(0, 3, 1, 2), x1_out.permute(0, 3, 1, 2)], done

    def _fpn_fused(self, x1, x2, x3_out, pk):
        sp = self.fpn_split
        o2, o1 = self.layer2_outconv2, self.layer1_outconv2
        # FPN merge (:108-109, :113-114): lateral 1x1 convolution + 2x bilinear upsampling of the coarser level, in the
        # convolution's epilogue when the level is even-sized and at least 32 wide, else conv + K8
        def merge(name, conv, fine, coarse):
            pc = pk.get(name, conv, None, sp)
            N, H, W, _ = fine.shape
            if H % 2 == 0 and W % 2 == 0 and W >= 32 and tuple(coarse.shape[1:3]) == (H // 2, W // 2):
                return ops.conv_nhwc(fine, pc, up=coarse)
            return ops.upsample2x_add(coarse.permute(0, 3, 1, 2), ops.conv_nhwc(fine, pc).permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        y = merge('layer2_outconv', self.layer2_outconv, x2, x3_out)
        y = ops.conv_nhwc(y, pk.get('o2.0', o2[0], o2[1], sp), act='leaky', slope=o2[2].negative_slope)
        x2_out = ops.conv_nhwc(y, pk.get('o2.3', o2[3], None, sp))
        y = merge('layer1_outconv', self.layer1_outconv, x1, x2_out)
        y = ops.conv_nhwc(y, pk.get('o1.0', o1[0], o1[1], sp), act='leaky', slope=o1[2].negative_slope)
        return ops.conv_nhwc(y, pk.get('o1.3', o1[3], None, sp))

    def forward(self, x, side=None):
        """side (inference on the GPU only): a HIP stream for the FPN's fine branch; the call then returns ([coarse, fine], event) and
        `fine` may be read only behind the event."""
        if _fused_ok(self, x) and x.shape[1] == 1:
            return self._forward_fused(x, side)
        if side is not None:
            raise ValueError('a side stream is a mode of the fused inference path')
        if not x.is_cuda or not ResNetFPN_8_2.hip_training:
            _vendor.require('the backbone on CPU tensors or with hip_training = False (vendor convolutions)')
        if (ResNetFPN_8_2.hip_training and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and x.shape[1] == 1
                and self.conv1.weight.requires_grad and self.conv1.out_channels in (64, 128)):
            x0 = ops.stem_train(x, self.conv1.weight)                                # K10 + its weight gradient
            x0 = ops.bn_act_train(x0, self.bn1, 'relu') if _bn_hip(self.bn1, x0) else self.relu(self.bn1(x0))
        else:
            x0 = self.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x3_out = _conv(self.layer3_outconv, x3, self)
        return [x3_out, self._fpn_plain(x1, x2, x3_out)]


def build_backbone(config):
    if config['backbone_type'] == 'ResNetFPN' and tuple(config['resolution']) == (8, 2):
        return ResNetFPN_8_2(config['resnetfpn'])
    raise ValueError(f"backbone {config['backbone_type']} / resolution {config['resolution']} not supported")
