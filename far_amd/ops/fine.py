"""far_amd.ops.fine: the fine level: K3 gather / scatter / expectation, K13 MLP block, K14 attention block (one family of the torch-tensor front ends for the C ABI in include/far_hip.h; far_amd/ops/__init__.py
re-exports everything under the flat far_amd.ops namespace the rest of the package uses)."""
import ctypes
import os
import threading

import torch

from .. import _lib, flags
from ._base import _CONV_ACT_EXP, _p, _stream, _written, overflow_flag


def fine_gather(feat_f, b_ids, cell_ids, wc, W, stride, out=None):
    """K3a.  feat_f: (N, C, Hf, Wf) fp32 in any strided layout (channels_last is the fast one).
    Returns (M, W*W, C): the windows F.unfold would have produced at the matched cells (written into `out` if given)."""
    lib = _lib.load()
    M = int(b_ids.shape[0])
    N, C, Hf, Wf = feat_f.shape
    if out is None:
        out = torch.empty(M, W * W, C, dtype=torch.float32, device=feat_f.device)
    elif tuple(out.shape) != (M, W * W, C) or not out.is_contiguous() or out.dtype != torch.float32:
        raise _lib.FarHipError('fine_gather: `out` must be a contiguous fp32 (M, W*W, C) tensor')
    if M == 0:
        return out
    if not feat_f.is_cuda or feat_f.dtype != torch.float32:
        raise _lib.FarHipError('fine_gather needs an fp32 GPU feature map')
    sn, sc, sh, sw = feat_f.stride()
    rc = lib.far_fine_gather_f32(ctypes.c_void_p(feat_f.data_ptr()), sn, sc, sh, sw, C, Hf, Wf,
                                 _p(b_ids, torch.int64), _p(cell_ids, torch.int64), int(wc), int(W), int(stride), M,
                                 _p(out), _stream())
    _lib.check(rc, 'far_fine_gather_f32')
    return _written(out)

DETERMINISTIC_FINE_SCATTER = True      # False: far_fine_scatter_f32 (fp32 atomics: the summation order varies from run to run)

class _FineWindowsFn(torch.autograd.Function):
    """K3a with its HIP backward: the M x 25 x C windows gathered directly (forward) and their gradients scattered back
    into the fine map (backward) -- the reference unfolds both full fine maps (123 MB per pair, fine_preprocess.py:40-44)
    and autograd folds them back."""

    @staticmethod
    def forward(ctx, feat_f, b_ids, cell_ids, wc, W, stride):
        f = feat_f.detach().float()
        out = fine_gather(f, b_ids, cell_ids, wc, W, stride)
        ctx.save_for_backward(b_ids, cell_ids)
        ctx.meta = (tuple(f.shape), tuple(f.stride()), int(wc), int(W), int(stride), feat_f.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        b_ids, cell_ids = ctx.saved_tensors
        shape, strides, wc, W, stride, dt = ctx.meta
        N, C, Hf, Wf = shape
        d = torch.empty_strided(shape, strides, dtype=torch.float32, device=g.device).zero_()
        M = int(b_ids.shape[0])
        if M:
            g = g.float().contiguous()
            hc = -(-Hf // stride)                                 # the coarse grid the cell ids index (Hf = stride * hc)
            if DETERMINISTIC_FINE_SCATTER and cell_ids.numel() and int(wc) * hc < (1 << 31) // max(N, 1):
                # fixed summation order: matches grouped by (image, cell) with a stable sort; one wave per fine-map pixel
                ncell = int(wc) * hc
                key = b_ids * ncell + cell_ids
                order = torch.argsort(key, stable=True)
                start = torch.zeros(N * ncell + 1, dtype=torch.int32, device=g.device)
                start[1:] = torch.cumsum(torch.bincount(key, minlength=N * ncell), 0).to(torch.int32)
                rc = lib.far_fine_scatter_det_f32(_p(g, torch.float32), strides[0], strides[1], strides[2], strides[3], C, Hf, Wf,
                                                  _p(order, torch.int64), _p(start, torch.int32), N, hc, wc, W, stride, M,
                                                  ctypes.c_void_p(d.data_ptr()), _stream())
                _lib.check(rc, 'far_fine_scatter_det_f32')
            else:
                rc = lib.far_fine_scatter_f32(_p(g, torch.float32), strides[0], strides[1], strides[2], strides[3], C, Hf, Wf,
                                              _p(b_ids, torch.int64), _p(cell_ids, torch.int64), wc, W, stride, M,
                                              ctypes.c_void_p(d.data_ptr()), _stream())
                _lib.check(rc, 'far_fine_scatter_f32')
        return d.to(dt), None, None, None, None, None

def fine_windows_train(feat_f, b_ids, cell_ids, wc, W, stride):
    """K3a, differentiable w.r.t. feat_f (N, C, Hf, Wf): (M, W*W, C) windows at the matched cells."""
    if not feat_f.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    if not (feat_f.is_contiguous() or feat_f.is_contiguous(memory_format=torch.channels_last)):
        feat_f = feat_f.contiguous()
    return _FineWindowsFn.apply(feat_f, b_ids.to(torch.int64).contiguous(), cell_ids.to(torch.int64).contiguous(), wc, W, stride)

def fine_expect(feat0, feat1, mkpts1_c, win_scale, scale1=None, b_ids=None):
    """K3b.  feat0/feat1: (M, WW, C).  Returns expec_f (M, 3), mkpts1_f (M, 2)."""
    lib = _lib.load()
    M, WW, C = feat0.shape
    W = int(round(WW ** 0.5))
    expec = torch.empty(M, 3, dtype=torch.float32, device=feat0.device)
    mk1 = torch.empty(M, 2, dtype=torch.float32, device=feat0.device)
    if M == 0:
        return expec, mk1
    rc = lib.far_fine_expect_f32(_p(feat0, torch.float32), _p(feat1, torch.float32), M, W, C,
                                 _p(mkpts1_c, torch.float32), float(win_scale), _p(scale1, torch.float32),
                                 _p(b_ids, torch.int64), _p(expec), _p(mk1), _stream())
    _lib.check(rc, 'far_fine_expect_f32')
    return expec, mk1

class PackedMlp:
    """Weight image of far_mlp_fused_f16s (K13): mlp[0] (2d x 2d) and mlp[2] (d x 2d) of a LoFTR encoder layer at d = 128 as
    24 slabs of 16 KiB in execution order, fp16 (hi, lo) planes, each tensor scaled by a power of two taken from its maximum
    (as PackedConv).  Slab s < 16 (k-step s of GEMM 1): [hidden tile t][plane][lane][8]: lane = (hidden channel 32 t + (lane & 31),
    half h = lane >> 5), element e = input channel 32 (s >> 1) + 16 h + 8 (s & 1) + e.  Slab 16 + t (hidden tile t of GEMM 2):
    [k-step u][output tile ct][plane][lane][8]: lane = (output channel 32 ct + (lane & 31), h), element e = hidden channel
    32 t + 16 u + 4 h + (e & 3) + 8 (e >> 2) -- the order in which GEMM 1's accumulator registers hold a row's hidden values."""

    def __init__(self, w0, w2):
        lib = _lib.load()
        w0, w2 = w0.detach().float(), w2.detach().float()
        d = w2.shape[0]
        if tuple(w0.shape) != (2 * d, 2 * d) or tuple(w2.shape) != (d, 2 * d) or lib.far_mlp_fused_packed_bytes(d) == 0:
            raise _lib.FarHipError(f'far_mlp_fused_f16s is built for d_model = 128 (got weights {tuple(w0.shape)}, {tuple(w2.shape)})')
        dev = w0.device

        def exp_of(w):
            amax = float(w.abs().max())                   # host sync at pack time only
            return 14 - (torch.frexp(torch.tensor(amax)).exponent.item() if amax > 0 else 0)

        def planes(v):                                    # (...,) fp32 (already scaled) -> (2, ...) fp16 hi / lo
            hi = v.half()
            return torch.stack([hi, (v - hi.float()).half()])

        self.e0, self.e2 = exp_of(w0), exp_of(w2)
        ar = lambda n: torch.arange(n, device=dev)
        s_, t_, l_, e_ = ar(16).view(16, 1, 1, 1), ar(8).view(1, 8, 1, 1), ar(64).view(1, 1, 64, 1), ar(8).view(1, 1, 1, 8)
        hc = 32 * t_ + (l_ & 31)
        k = 32 * (s_ >> 1) + 16 * (l_ >> 5) + 8 * (s_ & 1) + e_
        g1 = planes(w0[hc.expand(16, 8, 64, 8), k.expand(16, 8, 64, 8)] * 2.0 ** self.e0)         # (2, s, t, l, e)
        g1 = g1.permute(1, 2, 0, 3, 4).contiguous()                                               # (s, t, plane, l, e)
        t2, u2, c2, l2, e2 = (ar(8).view(8, 1, 1, 1, 1), ar(2).view(1, 2, 1, 1, 1), ar(4).view(1, 1, 4, 1, 1),
                              ar(64).view(1, 1, 1, 64, 1), ar(8).view(1, 1, 1, 1, 8))
        co = 32 * c2 + (l2 & 31)
        hk = 32 * t2 + 16 * u2 + 4 * (l2 >> 5) + (e2 & 3) + 8 * (e2 >> 2)
        shp = (8, 2, 4, 64, 8)
        g2 = planes(w2[co.expand(shp), hk.expand(shp)] * 2.0 ** self.e2)                          # (2, t, u, ct, l, e)
        g2 = g2.permute(1, 2, 3, 0, 4, 5).contiguous()                                            # (t, u, ct, plane, l, e)
        self.packed = torch.cat([g1.reshape(-1), g2.reshape(-1)]).view(torch.uint8)
        assert self.packed.numel() == lib.far_mlp_fused_packed_bytes(d)
        self.d = d
        self.hscale = 2.0 ** -self.e0                     # accumulator of GEMM 1 -> 2^4 x hidden
        self.oscale = 2.0 ** -(self.e2 + _CONV_ACT_EXP)   # accumulator of GEMM 2 -> output

def mlp_fused(x, msg, pack, gamma, beta, eps, out=None, plain16=False):
    """K13: x + LayerNorm(W2 relu(W0 [x | msg])) for (.., 128) fp32 tensors (transformer.py:64-67 at d_model = 128).
    plain16: plain fp16 operands (far_mlp_fused_f16) instead of split pairs."""
    lib = _lib.load()
    if x.shape != msg.shape or x.shape[-1] != pack.d:
        raise _lib.FarHipError('mlp_fused: x and msg must both be (..., 128)')
    R = x.numel() // pack.d
    y = torch.empty_like(x) if out is None else out
    name = 'far_mlp_fused_f16' if plain16 else 'far_mlp_fused_f16s'
    rc = getattr(lib, name)(_p(x, torch.float32), _p(msg, torch.float32), _p(pack.packed), R, pack.d, pack.hscale, pack.oscale,
                            _p(gamma, torch.float32), _p(beta, torch.float32), float(eps), _p(y, torch.float32),
                            _p(overflow_flag(x.device)), _stream())
    _lib.check(rc, name)
    return y if out is None else _written(y)

class PackedAttn:
    """Weight image of far_attn_block_f16s (K14): q / k / v / merge projections (128 x 128, no bias) of a LoFTR encoder layer
    as 16 slabs of 16 KiB in execution order [k c0][v c0] .. [k c3][v c3][q c0..c3][merge t0..t3], fp16 (hi, lo) planes, each
    tensor scaled by a power of two from its maximum.  Projection slab (chunk c of W): [k-step ks][tile t][plane][lane][8]:
    lane = (output channel 32 t + (lane & 31), h = lane >> 5), element e = input channel 32 c + 16 h + 8 ks + e.  Merge slab t
    (as PackedMlp's second half): [k-step u][output tile ct][plane][lane][8], element e = input channel
    32 t + 16 u + 4 h + (e & 3) + 8 (e >> 2)."""

    def __init__(self, wq, wk, wv, wm):
        lib = _lib.load()
        ws = [w.detach().float() for w in (wk, wv, wq, wm)]
        d = ws[0].shape[0]
        if any(tuple(w.shape) != (d, d) for w in ws) or lib.far_attn_block_packed_bytes(d) == 0:
            raise _lib.FarHipError('far_attn_block_f16s is built for d_model = 128')
        dev = ws[0].device

        def exp_of(w):
            amax = float(w.abs().max())
            return 14 - (torch.frexp(torch.tensor(amax)).exponent.item() if amax > 0 else 0)

        def planes(v):
            hi = v.half()
            return torch.stack([hi, (v - hi.float()).half()])

        self.exps = [exp_of(w) for w in ws]
        ar = lambda n: torch.arange(n, device=dev)
        c_, k_, t_, l_, e_ = (ar(4).view(4, 1, 1, 1, 1), ar(2).view(1, 2, 1, 1, 1), ar(4).view(1, 1, 4, 1, 1),
                              ar(64).view(1, 1, 1, 64, 1), ar(8).view(1, 1, 1, 1, 8))
        shp = (4, 2, 4, 64, 8)
        co = (32 * t_ + (l_ & 31)).expand(shp)
        ci = (32 * c_ + 16 * (l_ >> 5) + 8 * k_ + e_).expand(shp)

        def proj(w, ex):                                   # -> (chunk, ks, tile, plane, lane, e)
            return planes(w[co, ci] * 2.0 ** ex).permute(1, 2, 3, 0, 4, 5).contiguous()

        pk, pv, pq = proj(ws[0], self.exps[0]), proj(ws[1], self.exps[1]), proj(ws[2], self.exps[2])
        kvi = torch.stack([pk, pv], 1).reshape(-1)          # [c][k | v][...]: slabs k c0, v c0, k c1, ...
        mi = (32 * c_ + 16 * k_ + 4 * (l_ >> 5) + (e_ & 3) + 8 * (e_ >> 2)).expand(shp)     # c_ = input tile t, k_ = u, t_ = output tile
        pm = planes(ws[3][co, mi] * 2.0 ** self.exps[3]).permute(1, 2, 3, 0, 4, 5).contiguous()
        self.packed = torch.cat([kvi, pq.reshape(-1), pm.reshape(-1)]).view(torch.uint8)
        assert self.packed.numel() == lib.far_attn_block_packed_bytes(d)
        self.d = d
        self.scales = [2.0 ** -(ex + _CONV_ACT_EXP) for ex in self.exps]       # k, v, q, merge

def attn_block(x, source, pack, nhead, gamma, beta, ln_eps, attn_eps=1e-6, out=None, plain16=False):
    """K14: norm1(merge(LinearAttention(q_proj(x), k_proj(source), v_proj(source)))) for (N, L <= 32, 128) windows
    (transformer.py:51-61 at d_model = 128, 8 heads).  plain16: plain fp16 operands (far_attn_block_f16)."""
    lib = _lib.load()
    N, L, d = x.shape
    S = source.shape[1]
    if d != pack.d or source.shape[0] != N or source.shape[2] != d:
        raise _lib.FarHipError('attn_block: x (N, L, 128) and source (N, S, 128) expected')
    y = torch.empty_like(x) if out is None else out
    sk, sv, sq, sm = pack.scales
    name = 'far_attn_block_f16' if plain16 else 'far_attn_block_f16s'
    rc = getattr(lib, name)(_p(x, torch.float32), _p(source, torch.float32), _p(pack.packed), N, L, S, d, int(nhead), sk, sv, sq, sm,
                            float(attn_eps), _p(gamma, torch.float32), _p(beta, torch.float32), float(ln_eps), _p(y, torch.float32),
                            _p(overflow_flag(x.device)), _stream())
    _lib.check(rc, name)
    return y if out is None else _written(y)
