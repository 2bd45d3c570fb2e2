"""far_amd.ops.conv: K9 / K17 / K10 convolutions (inference + training forms), BatchNorm K19, FPN upsample-add K8, affine epilogues K7 (one family of the torch-tensor front ends for the C ABI in include/far_hip.h; far_amd/ops/__init__.py
re-exports everything under the flat far_amd.ops namespace the rest of the package uses)."""
import ctypes
import os
import threading

import torch

from .. import _lib, flags
from ._base import _ACT, _p, _same_layout, _stream, _written, activation_exponent_value, grad_scale, overflow_flag
from .packs import PackedConv, WINO_MIN_ACT_EXP, WINO_MIN_PIXELS


class _ConvF16sFn(torch.autograd.Function):
    """A bias-free 3x3 / 1x1 convolution (stride 1 or 2, 'same' padding) of the ResNet-FPN backbone with gradients
    (resnet_fpn.py:5-12 conv1x1 / conv3x3 under autograd).  Forward: K9.  dgrad: K9 again -- the transposed convolution of a
    'same' stride-1 convolution is a 'same' stride-1 convolution with the spatially flipped, channel-transposed kernel; for
    stride 2 the output gradient is first spread onto the input grid (zeros in between).  wgrad: far_conv_wgrad_f32 when the
    library has it for the shape, else the vendor's backward-weights.  x, y: (N, C, H, W) logical, channels_last memory."""

    @staticmethod
    def forward(ctx, x, weight, stride, pack, pack_d):
        ks = int(weight.shape[-1])
        xn = x.detach().float().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)      # NHWC view
        xin = xn if (ks == 3 or stride == 1) else xn[:, ::stride, ::stride].contiguous()                # 1x1 stride 2
        y = conv_nhwc(xin, pack())
        ctx.save_for_backward(xn, weight)
        ctx.stride, ctx.pack_d = int(stride), pack_d
        ctx.act_exp = activation_exponent_value()            # the weight gradient splits xn with the forward's exponent
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        xn, weight = ctx.saved_tensors
        ks, st = int(weight.shape[-1]), ctx.stride
        N, H, W, Cin = xn.shape
        gn = g.float().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)                # (N, Ho, Wo, Cout)
        dx = dw = None
        # gradients sit anywhere in magnitude: a power-of-two scale from the maximum places them in K9's window (as _LinearF16sFn)
        gs = gn.contiguous()
        sc = grad_scale(gs)
        if ctx.needs_input_grad[0]:
            if st == 1:
                dxn = conv_nhwc(gs, ctx.pack_d(), act_scale_dev=sc)
            elif ks == 3:
                up = torch.zeros(N, H, W, gs.shape[-1], dtype=torch.float32, device=g.device)
                up[:, ::st, ::st] = gs
                dxn = conv_nhwc(up, ctx.pack_d(), act_scale_dev=sc)
            else:
                dxn = torch.zeros(N, H, W, Cin, dtype=torch.float32, device=g.device)
                dxn[:, ::st, ::st] = conv_nhwc(gs, ctx.pack_d(), act_scale_dev=sc)
            dx = dxn.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(xn, gs, ks, st, dy_scale=sc, act_exp=ctx.act_exp)
            if dw is None:                                                                              # shape without a kernel: vendor
                dw = torch.ops.aten.convolution_backward(gn.permute(0, 3, 1, 2), xn.permute(0, 3, 1, 2), weight, None, [st, st],
                                                         [ks // 2, ks // 2], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw, None, None, None

def conv_wgrad(xn, gn, ks, stride, dy_scale=None, act_exp=None, keep=None):
    """K16.  dW (Cout, Cin, ks, ks) of a 'same' bias-free convolution from its NHWC input xn (N, H, W, Cin) and NHWC output
    gradient gn (N, Ho, Wo, Cout); split-fp16 operands, deterministic two-stage sum.  dy_scale = grad_scale(gn) if the caller has
    it already; act_exp: the activation exponent xn was consumed with in the forward (default: the current one).  None only on
    the comparison leg (USE_HIP_WGRAD False)."""
    lib = _lib.load()
    N, H, W, Cin = xn.shape
    Cout = gn.shape[-1]
    if not USE_HIP_WGRAD:
        return None
    xn, gn = xn.contiguous(), gn.contiguous()
    dw = torch.empty(Cout, Cin, ks, ks, dtype=torch.float32, device=xn.device)
    nb = int(lib.far_conv_wgrad_ws_bytes(N, H, W, Cin, Cout, ks, stride))
    ws = torch.empty(nb, dtype=torch.uint8, device=xn.device)
    rc = lib.far_conv_wgrad_f16s(_p(xn, torch.float32), _p(gn, torch.float32), N, H, W, Cin, Cout, ks, stride,
                                 activation_exponent_value() if act_exp is None else int(act_exp),
                                 _p(dy_scale) if dy_scale is not None else None, _p(ws), nb, _p(dw),
                                 overflow_flag(xn.device).data_ptr(), _stream())
    _lib.check(rc, 'far_conv_wgrad_f16s')
    if keep is not None:
        keep += [ws, xn, gn]                     # launched on a side stream: alive until the caller's join
    return dw

USE_HIP_WGRAD = True       # False: the vendor's backward-weights / GEMM (comparison leg of bench.py --workload c3 --vendor-train)

def linear_wgrad(x2, g2, dy_scale=None, act_exp=None, keep=None):
    """K16 as a Linear layer's weight gradient: dW (N_out, K) = g2^T x2 for x2 (rows, K), g2 (rows, N_out); None -> caller's GEMM
    (comparison leg only)."""
    rows, K = x2.shape
    if not USE_HIP_WGRAD or rows == 0:
        return None
    h = rows // 32 if rows % 32 == 0 else 1                     # 1x1 kernel: any factoring of the rows into H x W is the same sum
    return conv_wgrad(x2.reshape(1, h, rows // h, K), g2.reshape(1, h, rows // h, g2.shape[1]), 1, 1, dy_scale, act_exp, keep).reshape(g2.shape[1], K)

def conv_train(x, weight, stride, cache, name, split=True):
    """K9 convolution with gradients.  x (N, Cin, H, W) fp32 GPU; weight (Cout, Cin, k, k), k in {1, 3}; cache: a PackCache."""
    if not x.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    ks = int(weight.shape[-1])
    re = lambda pc: pc.refresh(weight)                 # after an optimizer step: same buffers, one launch (the forward image first:
    pack = lambda: cache.get((name, 'fwd', split), [weight], lambda: PackedConv(weight, split=split, stride=stride if ks == 3 else 1), refresh=re)
    pack_d = lambda: cache.get((name, 'dgrad', split), [weight],   # the dgrad image borrows its scale; the forward ran before the backward)
                               lambda: PackedConv(weight, split=split, dgrad=True, pack_scale=pack().pack_scale),
                               refresh=re).follow_scale(pack(), weight)
    return _ConvF16sFn.apply(x, weight, int(stride), pack, pack_d)

def affine_act(x, scale, shift, residual=None, act='relu', slope=0.01, inplace=True):
    """K7.  y = act(x * scale[c] + shift[c] (+ residual)) for x (N, C, H, W) in NCHW or channels_last memory."""
    lib = _lib.load()
    N, C, H, W = x.shape
    lay, ts = _same_layout([x] + ([residual] if residual is not None else []))
    x = ts[0]
    residual = ts[1] if residual is not None else None
    if (lay == 0 and (H * W) % 4) or (lay == 1 and C % 4):
        raise _lib.FarHipError('affine_act needs HW % 4 == 0 (NCHW) or C % 4 == 0 (channels_last)')
    y = x if inplace else torch.empty_like(x)
    code = {'none': 0, 'relu': 1, 'leaky': 2}[act]
    rc = lib.far_affine_act_f32(ctypes.c_void_p(x.data_ptr()), _p(scale, torch.float32), _p(shift, torch.float32),
                                ctypes.c_void_p(residual.data_ptr() if residual is not None else 0),
                                N, C, H * W, lay, code, float(slope), ctypes.c_void_p(y.data_ptr()), _stream())
    _lib.check(rc, 'far_affine_act_f32')
    return _written(y) if inplace else y

def upsample2x_add(lo, hi):
    """K8.  hi + F.interpolate(lo, scale_factor=2, mode='bilinear', align_corners=True)."""
    lib = _lib.load()
    N, C, h, w = lo.shape
    if tuple(hi.shape) != (N, C, 2 * h, 2 * w):
        raise _lib.FarHipError(f'upsample2x_add shape mismatch {tuple(lo.shape)} vs {tuple(hi.shape)}')
    lay, (hi, lo) = _same_layout([hi, lo])
    out = torch.empty_like(hi)
    rc = lib.far_upsample2x_add_f32(ctypes.c_void_p(lo.data_ptr()), ctypes.c_void_p(hi.data_ptr()), N, h, w, C, lay,
                                    ctypes.c_void_p(out.data_ptr()), _stream())
    _lib.check(rc, 'far_upsample2x_add_f32')
    return out

class _BatchNormActFn(torch.autograd.Function):
    """act(bn(x) (+ residual)) with BATCH statistics and gradients (resnet_fpn.py:24-41, 60-62, 75-91 under autograd, nn.BatchNorm2d
    in training mode): K19 statistics + K7 normalise / activate / add forward, far_bn_train_bwd_f32 backward; deterministic.
    Tensors (N, C, H, W) logical, channels_last memory.  The running statistics of `bn` are updated as the module does."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, act, slope, residual):
        lib = _lib.load()
        cl = torch.channels_last
        xc = x.detach().contiguous(memory_format=cl)
        N, C, H, W = xc.shape
        M = N * H * W
        nb = int(lib.far_bn_train_ws_bytes(M, C))
        buf = torch.empty(4 * C + nb // 4, dtype=torch.float32, device=xc.device)     # { scale, shift, mean, rstd } + partial sums
        y = torch.empty_like(xc)
        track = bn.track_running_stats and bn.running_mean is not None
        r = None if residual is None else residual.detach().contiguous(memory_format=cl)
        vp = buf.data_ptr()
        rc = lib.far_bn_act_train_fwd_f32(xc.data_ptr(), 0 if r is None else r.data_ptr(), M, C,
                                          0 if weight is None else weight.data_ptr(), 0 if bias is None else bias.data_ptr(), float(bn.eps),
                                          float(bn.momentum), bn.running_mean.data_ptr() if track else 0,
                                          bn.running_var.data_ptr() if track else 0, _ACT[act], float(slope), y.data_ptr(), vp,
                                          vp + 16 * C, nb, _stream())
        _lib.check(rc, 'far_bn_act_train_fwd_f32')
        if track:
            bn.num_batches_tracked += 1
            _written(bn.running_mean)                # K19 updated them through raw pointers: the inference image folded from the
            _written(bn.running_var)                 # old statistics (PackCache stamps on their versions) must not be reused
        ctx.save_for_backward(xc, y, buf, weight)
        ctx.act, ctx.slope, ctx.has_res, ctx.nb = _ACT[act], float(slope), residual is not None, nb
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        xc, y, buf, weight = ctx.saved_tensors
        N, C, H, W = xc.shape
        M = N * H * W
        gc = g.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(xc)
        dres = torch.empty_like(xc) if ctx.has_res else None
        dgb = torch.empty(2, C, dtype=torch.float32, device=xc.device)
        vp = buf.data_ptr()
        rc = lib.far_bn_train_bwd_f32(xc.data_ptr(), gc.data_ptr(), y.data_ptr(), vp + 8 * C, vp + 12 * C,
                                      0 if weight is None else weight.data_ptr(), M, C, ctx.act, ctx.slope, dx.data_ptr(), dgb.data_ptr(),
                                      dgb.data_ptr() + 4 * C, 0 if dres is None else dres.data_ptr(), vp + 16 * C, ctx.nb, _stream())
        _lib.check(rc, 'far_bn_train_bwd_f32')
        return (dx, dgb[0] if weight is not None else None, dgb[1] if weight is not None else None, None, None, None, dres)

USE_HIP_BATCHNORM_TRAIN = not flags.off('FAR_NO_BN')      # False: nn.BatchNorm2d + torch activations under autograd (comparison leg, --vendor-train)

def bn_act_train(x, bn, act='none', slope=0.01, residual=None):
    """K19.  act(bn(x) (+ residual)) for an nn.BatchNorm2d in TRAINING mode (batch statistics) on a GPU tensor, with gradients;
    what the kernels do not cover (eval-mode modules, momentum None, C % 4 != 0, C > 1024) runs the module and torch activations."""
    if not x.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    C = x.shape[1]
    if (not USE_HIP_BATCHNORM_TRAIN or type(bn) is not torch.nn.BatchNorm2d or not bn.training or bn.momentum is None or (C & 3) or C > 1024 or x.dtype != torch.float32
            or x.numel() == 0 or (bn.weight is None) != (bn.bias is None)):
        y = bn(x)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if act == 'relu' else (torch.nn.functional.leaky_relu(y, slope) if act == 'leaky' else y)
    return _BatchNormActFn.apply(x, bn.weight, bn.bias, bn, act, slope, residual)

class _Upsample2xAddFn(torch.autograd.Function):
    """hi + F.interpolate(lo, scale_factor=2, mode='bilinear', align_corners=True) with gradients (resnet_fpn.py:108-109,
    :113-114 under autograd): forward K8, backward far_upsample2x_bwd_f32 -- a gather in a fixed order, where the node torch
    records for F.interpolate scatters with atomics (the one source of run-to-run differences in the backbone gradients)."""

    @staticmethod
    def forward(ctx, lo, hi):
        cl = torch.channels_last
        return upsample2x_add(lo.detach().contiguous(memory_format=cl), hi.detach().contiguous(memory_format=cl))

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        N, C, H, W = g.shape
        dlo = None
        if ctx.needs_input_grad[0]:
            gn = g.float().contiguous(memory_format=torch.channels_last)
            dlo = torch.empty(N, C, H // 2, W // 2, dtype=torch.float32, device=g.device, memory_format=torch.channels_last)
            rc = lib.far_upsample2x_bwd_f32(ctypes.c_void_p(gn.data_ptr()), N, H // 2, W // 2, C, ctypes.c_void_p(dlo.data_ptr()),
                                            _stream())
            _lib.check(rc, 'far_upsample2x_bwd_f32')
        return dlo, (g if ctx.needs_input_grad[1] else None)

def upsample2x_add_train(lo, hi):
    """K8 with gradients; lo (N, C, h, w), hi (N, C, 2h, 2w) fp32 GPU, C % 4 == 0."""
    return _Upsample2xAddFn.apply(lo, hi)

USE_WINO = not flags.off('FAR_NO_WINO')      # inference 3x3 stride-1 layers on K17 (conv_nhwc dispatches); False: K9 everywhere

def conv3x3_wino(x, pw, residual=None, act='none', slope=0.01, out=None):
    """K17.  x (N, H, W, Cin) fp32 contiguous -> act(conv3x3(x) * scale + shift (+ residual)) as (N, H, W, Cout): conv_nhwc's
    result for a stride-1 3x3 layer, on the Winograd kernel."""
    lib = _lib.load()
    N, H, W, Cin = x.shape
    if Cin != pw.Cin:
        raise _lib.FarHipError(f'conv3x3_wino: input has {Cin} channels, weights expect {pw.Cin}')
    shape = (N, H, W, pw.Cout)
    if out is None:
        y = torch.empty(shape, dtype=torch.float32, device=x.device)
    else:
        y = out
        if y.numel() != N * H * W * pw.Cout or not y.is_contiguous() or y.dtype != torch.float32:
            raise _lib.FarHipError('conv3x3_wino: `out` must be a contiguous fp32 tensor of the output size')
    if residual is not None and (residual.numel() != y.numel() or not residual.is_contiguous()):
        raise _lib.FarHipError('conv3x3_wino: `residual` must be a contiguous tensor of the output size')
    ptr = lambda t: _p(t, torch.float32).value
    d = _lib.ConvDesc(x=ptr(x), x2=None, packed=_p(pw.packed).value, scale=ptr(pw.scale), shift=ptr(pw.shift), res=ptr(residual),
                      ln_gamma=None, ln_beta=None, post_res=None, up=None, y=ptr(y), N=N, H=H, W=W, Cin=Cin, Cin1=Cin, Cout=pw.Cout,
                      ksize=3, stride=1, act=_ACT[act], split=1, out_planes=1, res_group=1, slope=float(slope), ln_eps=0.0,
                      act_exp=max(activation_exponent_value(), 0), overflow=overflow_flag(x.device).data_ptr(), act_scale_dev=None)
    _lib.check(lib.far_conv3x3_wino_f32(ctypes.byref(d), _stream()), 'far_conv3x3_wino_f32')
    return y if out is None else _written(y)

def conv_nhwc(x, pc, residual=None, act='none', slope=0.01, x2=None, out_planes=1, res_group=1, ln=None,
              post_residual=None, out=None, up=None, act_scale_dev=None, in_stride=1):
    """K9.  x (N, H, W, Cin) fp32 contiguous -> act(conv(x) * scale + shift (+ residual)) as (N, H, W, Cout).
    With x2 (N, H, W, C2) the convolution input is cat([x, x2], -1), read in place.  out_planes = P > 1 returns
    (P, N, H, W, Cout / P): the output channels split into P separate contiguous tensors.
    up (N, H/2, W/2, Cout), 1x1 convolutions: + F.interpolate(up, scale_factor=2, bilinear, align_corners=True)
    in the epilogue (the FPN merge).  in_stride = 2 (1x1 weights): the convolution of x[:, ::2, ::2] read in place -- the
    down-sampling shortcut of a BasicBlock (resnet_fpn.py:26-29) without the subsampled copy."""
    lib = _lib.load()
    N, H, W, Cin1 = x.shape
    if in_stride not in (1, 2) or (in_stride == 2 and (pc.ksize != 1 or pc.stride != 1 or x2 is not None or up is not None or res_group != 1)):
        raise _lib.FarHipError('conv_nhwc: in_stride = 2 is the in-place subsampling of a plain 1x1 convolution')
    if (USE_WINO and pc.ksize == 3 and pc.stride == 1 and pc.split and x2 is None and out_planes == 1 and res_group == 1 and ln is None
            and post_residual is None and up is None and act_scale_dev is None and not torch.is_grad_enabled()
            and activation_exponent_value() >= WINO_MIN_ACT_EXP and H * W >= WINO_MIN_PIXELS and H * W * pc.Cout < 2 ** 31):
        pw = pc.wino()
        if pw is not None:
            # K17 (Winograd F(2x2, 3x3)): 1.03-1.26x K9 on the backbone's stride-1 3x3 layers at one third of its error
            return conv3x3_wino(x, pw, residual=residual, act=act, slope=slope, out=out)
    if up is not None and (tuple(up.shape) != (N, H // 2, W // 2, pc.Cout) or not up.is_contiguous()):
        raise _lib.FarHipError(f'conv_nhwc: `up` must be a contiguous ({N}, {H // 2}, {W // 2}, {pc.Cout}) tensor')
    Cin = Cin1 + (x2.shape[-1] if x2 is not None else 0)
    if Cin != pc.Cin or (x2 is not None and tuple(x2.shape[:3]) != (N, H, W)):
        raise _lib.FarHipError(f'conv_nhwc: input has {Cin} channels, weights expect {pc.Cin}')
    st = pc.stride * in_stride
    if pc.Cout % out_planes:
        raise _lib.FarHipError('conv_nhwc: out_planes must divide the output channel count')
    shape = (N, (H - 1) // st + 1, (W - 1) // st + 1, pc.Cout // out_planes)
    full = (out_planes,) + shape if out_planes > 1 else shape
    if out is None:
        y = torch.empty(full, dtype=torch.float32, device=x.device)
    else:
        y = out
        n_out = 1
        for d in full:
            n_out *= d
        if y.numel() != n_out or not y.is_contiguous() or y.dtype != torch.float32:
            raise _lib.FarHipError('conv_nhwc: `out` must be a contiguous fp32 tensor of the output size')
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    ptr = lambda t: _p(t, torch.float32).value
    d = _lib.ConvDesc(x=ptr(x), x2=ptr(x2), packed=_p(pc.packed).value, scale=ptr(pc.scale), shift=ptr(pc.shift),
                      res=ptr(residual), ln_gamma=ptr(g), ln_beta=ptr(b), post_res=ptr(post_residual), up=ptr(up), y=ptr(y),
                      N=N, H=H, W=W, Cin=Cin, Cin1=Cin1, Cout=pc.Cout, ksize=pc.ksize, stride=st, act=_ACT[act],
                      split=int(pc.split), out_planes=int(out_planes), res_group=int(res_group), slope=float(slope),
                      ln_eps=float(eps), act_exp=activation_exponent_value(), overflow=overflow_flag(x.device).data_ptr(),
                      act_scale_dev=None if act_scale_dev is None else act_scale_dev.data_ptr())
    rc = lib.far_conv_nhwc_f32(ctypes.byref(d), _stream())
    _lib.check(rc, 'far_conv_nhwc_f32')
    return y if out is None else _written(y)

def stem7x7(img, weight, scale=None, shift=None):
    """K10.  img (N, 1, H, W) fp32 -> relu(bn(conv7x7 stride 2)) as NHWC (N, H/2, W/2, Cout); without scale / shift the bare
    convolution (training: BatchNorm follows with batch statistics)."""
    lib = _lib.load()
    N, one, H, W = img.shape
    Cout = weight.shape[0]
    if one != 1 or tuple(weight.shape[1:]) != (1, 7, 7):
        raise _lib.FarHipError('stem7x7 expects a 1-channel image and a (Cout, 1, 7, 7) weight')
    y = torch.empty(N, (H + 1) // 2, (W + 1) // 2, Cout, dtype=torch.float32, device=img.device)
    rc = lib.far_stem7x7_nhwc_f32(_p(img.contiguous(), torch.float32), _p(weight.detach().contiguous(), torch.float32),
                                  _p(scale, torch.float32) if scale is not None else None,
                                  _p(shift, torch.float32) if shift is not None else None, N, H, W, Cout, _p(y), _stream())
    _lib.check(rc, 'far_stem7x7_nhwc_f32')
    return y

class _StemFn(torch.autograd.Function):
    """The stem convolution with its weight gradient (resnet_fpn.py:60 under autograd; the image needs no gradient): K10 forward
    without the BatchNorm fold, far_stem7x7_wgrad_f32 backward.  Returns (N, Cout, H/2, W/2) logical, channels_last memory."""

    @staticmethod
    def forward(ctx, img, weight):
        img = img.detach().float().contiguous()
        ctx.save_for_backward(img)
        return stem7x7(img, weight).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        img, = ctx.saved_tensors
        lib = _lib.load()
        N, _, H, W = img.shape
        gn = g.float().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).contiguous()
        Cout = gn.shape[-1]
        nb = int(lib.far_stem7x7_wgrad_ws_bytes(N, H, W, Cout))
        ws = torch.empty(nb, dtype=torch.uint8, device=img.device)
        dw = torch.empty(Cout, 1, 7, 7, dtype=torch.float32, device=img.device)
        rc = lib.far_stem7x7_wgrad_f32(_p(img, torch.float32), _p(gn, torch.float32), N, H, W, Cout, _p(ws), nb, _p(dw), _stream())
        _lib.check(rc, 'far_stem7x7_wgrad_f32')
        return None, dw

def stem_train(img, weight):
    """K10 with the weight gradient: img (N, 1, H, W) fp32 GPU, weight (Cout, 1, 7, 7) -> conv7x7 stride 2 (no BN, no ReLU)."""
    if not img.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    return _StemFn.apply(img, weight)
