"""far_amd.ops.attention: K5 linear attention and K6 LayerNorm (forward, backward, training wrappers) (one family of the torch-tensor front ends for the C ABI in include/far_hip.h; far_amd/ops/__init__.py
re-exports everything under the flat far_amd.ops namespace the rest of the package uses)."""
import ctypes
import os
import threading

import torch

from .. import _lib, flags
from ._base import _p, _stream, _written, _ws


def linear_attention(q, k, v, nhead, q_mask=None, kv_mask=None, eps=1e-6):
    """K5.  q: (N, L, C), k, v: (N, S, C) raw projections; returns (N, L, C) (heads concatenated)."""
    lib = _lib.load()
    N, L, C = q.shape
    S = k.shape[1]
    D = C // nhead
    out = torch.empty(N, L, C, dtype=torch.float32, device=q.device)
    if N == 0:
        return out
    ws = _ws(lib.far_linear_attention_workspace_bytes(N, S, nhead, D), q.device)
    rc = lib.far_linear_attention_f32(_p(q, torch.float32), _p(k, torch.float32), _p(v, torch.float32), N, L, S,
                                      nhead, D, _p(q_mask, torch.uint8), _p(kv_mask, torch.uint8), float(eps),
                                      _p(out), _p(ws), _stream())
    _lib.check(rc, 'far_linear_attention_f32')
    return out

class _LinearAttentionFn(torch.autograd.Function):
    """K5 with its HIP backward (far_linear_attention_f32 / far_linear_attention_bwd_f32)."""

    @staticmethod
    def forward(ctx, q, k, v, nhead, q_mask, kv_mask, eps):
        qc, kc, vc = (t.detach().float().contiguous() for t in (q, k, v))
        out = linear_attention(qc, kc, vc, nhead, q_mask, kv_mask, eps)
        ctx.save_for_backward(qc, kc, vc, q_mask if q_mask is not None else torch.empty(0), kv_mask if kv_mask is not None else torch.empty(0))
        ctx.nhead, ctx.eps, ctx.has = nhead, eps, (q_mask is not None, kv_mask is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        qc, kc, vc, qm, km = ctx.saved_tensors
        dq, dk, dv = linear_attention_bwd(qc, kc, vc, g, ctx.nhead, qm if ctx.has[0] else None, km if ctx.has[1] else None, ctx.eps)
        return dq, dk, dv, None, None, None, None

def linear_attention_bwd(qc, kc, vc, g, nhead, q_mask=None, kv_mask=None, eps=1e-6):
    """K5 backward: (dq, dk, dv) of linear_attention(qc, kc, vc) for the output gradient g; all (N, L | S, C) fp32 contiguous."""
    lib = _lib.load()
    N, L, C = qc.shape
    S = kc.shape[1]
    D = C // nhead
    g = g.float().contiguous()
    dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
    if N:
        ws = _ws(lib.far_linear_attention_bwd_workspace_bytes(N, L, S, nhead, D), qc.device)
        rc = lib.far_linear_attention_bwd_f32(_p(qc), _p(kc), _p(vc), _p(g, torch.float32), N, L, S, nhead, D,
                                              _p(q_mask, torch.uint8), _p(kv_mask, torch.uint8), float(eps), _p(dq), _p(dk), _p(dv),
                                              _p(ws), _stream())
        _lib.check(rc, 'far_linear_attention_bwd_f32')
    return dq, dk, dv

def linear_attention_train(q, k, v, nhead, q_mask=None, kv_mask=None, eps=1e-6):
    """K5 with gradients: q (N, L, C), k, v (N, S, C) raw projections -> (N, L, C)."""
    as_u8 = lambda m: None if m is None else m.to(torch.uint8).contiguous()
    return _LinearAttentionFn.apply(q, k, v, nhead, as_u8(q_mask), as_u8(kv_mask), eps)

def layernorm(x, weight, bias, eps=1e-5, residual=None, out=None):
    """K6.  LayerNorm over the last dim (+ residual).  x: (..., C) fp32 contiguous GPU tensor; `out`: optional
    contiguous destination of the same shape (e.g. one half of a buffer that a later stage wants concatenated)."""
    lib = _lib.load()
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x) if out is None else out
    if y.shape != x.shape:
        raise _lib.FarHipError('layernorm: `out` must have the shape of x')
    rc = lib.far_layernorm_f32(_p(x, torch.float32), _p(weight, torch.float32), _p(bias, torch.float32),
                               _p(residual, torch.float32), rows, C, float(eps), _p(y), _stream())
    _lib.check(rc, 'far_layernorm_f32')
    return y if out is None else _written(y)

class _LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension (+ residual) with gradients (transformer.py:61, 65-67 under autograd): K6 forward,
    far_layernorm_bwd_f32 backward (dx; dgamma / dbeta summed in a fixed order).  The residual's gradient is dy itself."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, residual):
        xc = x.detach().float().contiguous()
        r = None if residual is None else residual.detach().float().contiguous()
        y = layernorm(xc, weight.detach(), bias.detach(), eps, residual=r)
        ctx.save_for_backward(xc, weight)
        ctx.eps, ctx.has_res = float(eps), residual is not None
        return y

    @staticmethod
    def backward(ctx, g):
        xc, weight = ctx.saved_tensors
        g = g.float().contiguous()
        dx, dg, db = layernorm_bwd(xc, weight, g, ctx.eps)
        return dx, dg, db, None, (g if ctx.has_res else None)

def layernorm_bwd(xc, weight, g, eps):
    """K6 backward: (dx, dgamma, dbeta) of LayerNorm(xc; eps) * weight + bias over the last dimension for the output gradient g
    (fp32 contiguous, C % 4 == 0, C <= 1024)."""
    lib = _lib.load()
    C = xc.shape[-1]
    rows = xc.numel() // C
    nb = int(lib.far_layernorm_bwd_ws_bytes(rows, C))
    if nb == 0:
        raise _lib.FarHipError(f'layernorm_bwd: {C} channels not covered (C % 4 == 0, C <= 1024)')
    ws = torch.empty(nb, dtype=torch.uint8, device=xc.device)
    dx = torch.empty_like(xc)
    dgb = torch.empty(2, C, dtype=torch.float32, device=xc.device)
    rc = lib.far_layernorm_bwd_f32(_p(xc, torch.float32), _p(weight.detach().contiguous(), torch.float32), _p(g, torch.float32), rows, C,
                                   float(eps), _p(dx), _p(dgb[0]), _p(dgb[1]), _p(ws), nb, _stream())
    _lib.check(rc, 'far_layernorm_bwd_f32')
    return dx, dgb[0], dgb[1]

USE_HIP_LAYERNORM_TRAIN = True      # False: nn.LayerNorm under autograd (comparison leg of bench.py --workload c3 --vendor-train)

def layernorm_train(x, norm, residual=None):
    """K6 with gradients: norm(x) (+ residual) for an nn.LayerNorm over the last dimension of a GPU tensor; shapes the backward
    kernel does not cover (C % 4 != 0 or C > 1024) use the module itself."""
    C = x.shape[-1]
    if not x.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    if (C & 3) or C > 1024 or norm.weight is None or norm.bias is None or x.numel() == 0 or not USE_HIP_LAYERNORM_TRAIN:
        y = norm(x)
        return y if residual is None else y + residual
    return _LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps, residual)

def linear_attention_apply(q, kv, nhead, S, q_mask=None, eps=1e-6):
    """The second half of K5: q (N, L, nhead * 32) raw projection and the state kv (N, nhead * 32, 33) -> (N, L, nhead * 32)."""
    lib = _lib.load()
    N, L, C = q.shape
    if C != nhead * 32 or tuple(kv.shape) != (N, C, 33):
        raise _lib.FarHipError('linear_attention_apply: q (N, L, nhead * 32), kv (N, nhead * 32, 33)')
    out = torch.empty(N, L, C, dtype=torch.float32, device=q.device)
    if N == 0:
        return out
    rc = lib.far_linear_attention_apply_f32(_p(q, torch.float32), _p(kv, torch.float32), N, L, int(S), nhead, _p(q_mask, torch.uint8),
                                            float(eps), _p(out), _stream())
    _lib.check(rc, 'far_linear_attention_apply_f32')
    return out
