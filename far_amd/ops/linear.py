"""far_amd.ops.linear: K9 in Linear mode: plain, gather, k|v-state, q-apply; the training Function (one family of the torch-tensor front ends for the C ABI in include/far_hip.h; far_amd/ops/__init__.py
re-exports everything under the flat far_amd.ops namespace the rest of the package uses)."""
import ctypes
import os
import threading

import torch

from .. import _lib, flags
from ._base import _ACT, _p, _stream, _ws, activation_exponent_value, grad_scale, overflow_flag
from .packs import train_pack, train_pack_t
from .conv import conv_nhwc, linear_wgrad


class _LinearF16sFn(torch.autograd.Function):
    """y = x W^T (+ b) on K9 with gradients: dx = dy W is K9 again (the weight packed transposed: a Linear layer whose
    weight is W^T), dW = dy^T x is K16 (the 1x1 case of the convolution weight gradient), db = column sums."""

    @staticmethod
    def forward(ctx, x, weight, bias, pack, pack_t):
        xc = x.detach().float().contiguous()
        y = linear_f16s(xc, pack())
        ctx.save_for_backward(xc, weight)
        ctx.pack_t, ctx.has_bias = pack_t, bias is not None
        ctx.act_exp = activation_exponent_value()            # the weight gradient splits xc with the forward's exponent
        return y

    @staticmethod
    def backward(ctx, g):
        xc, weight = ctx.saved_tensors
        g = g.float().contiguous()
        dx = None
        sc = grad_scale(g)
        if ctx.needs_input_grad[0]:
            # K9 splits its input into fp16 (hi, lo) pairs after a fixed 2^4 scale: fp32-grade for values in
            # ~[8e-3, 4e3], the range of activations -- gradients can sit anywhere (1e-7 is usual).  A power-of-two scale
            # taken from the tensor's maximum (on the device, no host sync) places them at the top of that window;
            # entries below max * 2^-17 keep 11 bits, which is 2^-28 of the maximum.
            # (far_grad_scale_f32 picks it on the device and K9 applies it inside the launch: no scaling passes over g / dx)
            dx = linear_f16s(g, ctx.pack_t(), act_scale_dev=sc)
        g2, x2 = g.reshape(-1, g.shape[-1]), xc.reshape(-1, xc.shape[-1])
        dw = None
        if ctx.needs_input_grad[1]:
            dw = linear_wgrad(x2.contiguous(), g2.contiguous(), sc, ctx.act_exp)
            if dw is None:
                dw = g2.t().mm(x2)
        db = g2.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None

def linear_train(x, weight, bias, cache, name, split=True):
    """K9 Linear with gradients.  cache: a PackCache; name: key prefix of this layer's forward / transposed weight images."""
    pack = lambda: train_pack(cache, name, weight, bias, split)
    pack_t = lambda: train_pack_t(cache, name, weight, bias, split)
    return _LinearF16sFn.apply(x, weight, bias, pack, pack_t)

def linear_f16s(x, pc, residual=None, act='none', x2=None, out_planes=1, res_group=1, ln=None, post_residual=None,
                out=None, act_scale_dev=None):
    """K9 as a linear layer: x (..., K) fp32 -> act(cat([x, x2], -1) W^T * scale + shift (+ residual)) (..., Cout);
    out_planes = P > 1: (P, ..., Cout / P), e.g. the q / k / v projections of one input in one launch.
    res_group = G > 1: residual is (rows / G, Cout), one row shared by each group of G consecutive rows.
    ln = (gamma, beta, eps): LayerNorm over the output channels fused into the epilogue (Cout 128 or 256), then
    + post_residual; out: optional destination."""
    lead = x.shape[:-1]
    rows = 1
    for d in lead:
        rows *= d
    r = None if residual is None else residual.reshape(1, 1, rows // res_group, pc.Cout)
    x2 = None if x2 is None else x2.reshape(1, 1, rows, x2.shape[-1])
    pr = None if post_residual is None else post_residual.reshape(1, 1, rows, pc.Cout)
    y = conv_nhwc(x.reshape(1, 1, rows, x.shape[-1]), pc, residual=r, act=act, x2=x2, out_planes=out_planes,
                  res_group=res_group, ln=ln, post_residual=pr, out=out, act_scale_dev=act_scale_dev)
    return y.reshape(*lead, pc.Cout) if out_planes == 1 else y.reshape(out_planes, *lead, pc.Cout // out_planes)

def kv_interleaved_weight(wk, wv, nhead):
    """The weight image far_linear_kv_f16s expects: the rows of Wk and Wv (each (H * 32, K)) head by head -- 64 j + [0, 32) = Wk's
    rows of head j, 64 j + [32, 64) = Wv's."""
    C, K = wk.shape
    if wv.shape != wk.shape or C != nhead * 32:
        raise _lib.FarHipError('kv_interleaved_weight: Wk, Wv must be (nhead * 32, K)')
    return torch.stack([wk.reshape(nhead, 32, K), wv.reshape(nhead, 32, K)], 1).reshape(2 * C, K)

def _linear_desc(x, pc, rows, y, out_planes, residual=None, res_group=1, act='none'):
    ptr = lambda t: _p(t, torch.float32).value
    return _lib.ConvDesc(x=ptr(x), x2=None, packed=_p(pc.packed).value, scale=ptr(pc.scale), shift=ptr(pc.shift), res=ptr(residual),
                         ln_gamma=None, ln_beta=None, post_res=None, up=None, y=ptr(y), N=1, H=1, W=rows, Cin=pc.Cin, Cin1=pc.Cin,
                         Cout=pc.Cout, ksize=1, stride=1, act=_ACT[act], split=int(pc.split), out_planes=out_planes, res_group=int(res_group),
                         slope=0.0, ln_eps=0.0, act_exp=activation_exponent_value(), overflow=overflow_flag(x.device).data_ptr(),
                         act_scale_dev=None)

def linear_gather_f16s(fmap, b_ids, cell_ids, wc, W, stride, pc, residual=None, res_group=1, act='none'):
    """K9 reading its rows through K3's window indices (far_linear_gather_f16s): fmap (n_img, Hf, Wf, C) fp32 NHWC contiguous,
    b_ids / cell_ids (M,) int64 -> act(windows W^T * scale + shift (+ residual)) as (M, W * W, Cout), where `windows` =
    fine_gather(fmap, b_ids, cell_ids, wc, W, stride) is never stored.  residual / res_group as linear_f16s."""
    lib = _lib.load()
    n_img, Hf, Wf, C = fmap.shape
    M = int(b_ids.shape[0])
    if not fmap.is_contiguous() or fmap.dtype != torch.float32 or C != pc.Cin or pc.ksize != 1 or not pc.split:
        raise _lib.FarHipError('linear_gather_f16s: needs a contiguous fp32 NHWC map and a split-operand Linear image of its channel count')
    rows = M * W * W
    out = torch.empty(M, W * W, pc.Cout, dtype=torch.float32, device=fmap.device)
    if rows:
        d = _linear_desc(fmap, pc, rows, out, 1, residual=residual, res_group=res_group, act=act)
        rc = lib.far_linear_gather_f16s(ctypes.byref(d), _p(b_ids, torch.int64), _p(cell_ids, torch.int64), int(wc), int(W), int(stride),
                                        int(n_img), int(Hf), int(Wf), _stream())
        _lib.check(rc, 'far_linear_gather_f16s')
    return out

def linear_kv_state(x, pc, S, want_image=False):
    """K9 + the K'^T V epilogue (far_linear_kv_f16s).  x (..., K) fp32 = n_img * S tokens, image after image; pc = PackedConv of
    kv_interleaved_weight(Wk, Wv, 8).  Returns the LinearAttention state (n_img, 256, 33) of linear_attention.py:38-45 --
    K'^T (V / S) per head and, in the last column, the sum of K' -- without k or v ever reaching memory; with want_image also the
    same state as the operand image linear_q_apply reads: (kv, image)."""
    lib = _lib.load()
    rows = 1
    for d in x.shape[:-1]:
        rows *= d
    if pc.Cout != 512 or pc.ksize != 1 or S < 64 or rows % S or x.shape[-1] != pc.Cin:
        raise _lib.FarHipError('linear_kv_state: needs a 512-row k | v weight image and whole images of S >= 64 tokens')
    kv = torch.empty(rows // S, 256, 33, dtype=torch.float32, device=x.device)
    img = torch.empty(int(lib.far_linear_kv_image_bytes(rows // S)), dtype=torch.uint8, device=x.device) if want_image else None
    if rows:
        ws = _ws(lib.far_linear_kv_workspace_bytes(rows, S), x.device)
        d = _linear_desc(x, pc, rows, None, 2)
        rc = lib.far_linear_kv_f16s(ctypes.byref(d), int(S), _p(ws), _p(kv), None if img is None else _p(img), _stream())
        _lib.check(rc, 'far_linear_kv_f16s')
    return (kv, img) if want_image else kv

def linear_q_apply(x, pc, image, S, eps=1e-6):
    """K9 + LinearAttention's second half in the epilogue (far_linear_q_apply_f16s).  x (N, L, K) fp32 query-side tokens, pc =
    PackedConv(Wq), image = linear_kv_state(source, ..., want_image=True)[1] of the N source images (S tokens each) -> the attention
    message (N, L, 256); q is never stored.  L >= 64."""
    lib = _lib.load()
    N, L, K = x.shape
    if pc.Cout != 256 or pc.ksize != 1 or K != pc.Cin or L < 64 or image.numel() != lib.far_linear_kv_image_bytes(N):
        raise _lib.FarHipError('linear_q_apply: needs a 256-row Wq image, L >= 64 and the state image of N source images')
    out = torch.empty(N, L, 256, dtype=torch.float32, device=x.device)
    if N:
        d = _linear_desc(x, pc, N * L, out, 1)
        rc = lib.far_linear_q_apply_f16s(ctypes.byref(d), int(L), int(S), _p(image), float(eps), _stream())
        _lib.check(rc, 'far_linear_q_apply_f16s')
    return out
