"""Thin torch-tensor front ends for the C ABI in include/far_hip.h.

PyTorch supplies device memory and the current HIP stream; all arithmetic of these ops happens in
libfar_hip.so.  Every op raises on CPU tensors -- there is no eager fallback.
"""
import os
import ctypes
import threading

import torch

from . import _lib, flags


_TLS = threading.local()          # .side: the library side stream this thread's launches currently go to (ops.side), or absent


def _stream():
    s = getattr(_TLS, 'side', None)
    return s if s is not None else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class side:
    """`with ops.side(i):` -- the launches inside go to the library's side stream i, which first waits for everything the
    current torch stream holds (far_stream_fork); they overlap with what the caller launches afterwards until ops.join(i).
    torch's allocator knows only the current stream, so every tensor the side launches touch must outlive the join: locals of
    the caller do; temporaries of the ops called inside are appended to a `keep` list by those ops (their `keep=` argument)."""

    def __init__(self, i):
        self.i = i

    def __enter__(self):
        h = _lib.load().far_stream_fork(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), self.i)
        if not h:
            raise _lib.FarHipError('far_stream_fork failed')
        self.prev = getattr(_TLS, 'side', None)
        _TLS.side = ctypes.c_void_p(h)

    def __exit__(self, *exc):
        _TLS.side = self.prev
        return False


def join(i):
    """The current torch stream waits for side stream i (far_stream_join)."""
    _lib.check(_lib.load().far_stream_join(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), i), 'far_stream_join')


def _p(t, dtype=None):
    """Device pointer of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    if not t.is_contiguous():
        raise _lib.FarHipError('far_amd ops need contiguous tensors')
    if dtype is not None and t.dtype != dtype:
        raise _lib.FarHipError(f'expected dtype {dtype}, got {t.dtype}')
    return ctypes.c_void_p(t.data_ptr())


def _written(t):
    """Tell autograd's version counter that `t` was overwritten through its raw device pointer (the kernels write
    caller-provided `out=` tensors behind torch's back; anything keyed on Tensor._version -- PackCache, the head's
    feature reuse -- must see it).  Inference tensors carry no version counter."""
    if t is not None and not t.is_inference():
        torch.autograd.graph.increment_version(t)
    return t


def tensor_version(t):
    """Tensor._version, or None for inference tensors (created under torch.inference_mode(): immutable outside it,
    no counter to read)."""
    return None if t.is_inference() else t._version


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def dual_softmax_stats(f0, f1, feat_div=1.0, sim_div=1.0, sim_mul=1.0, mask0=None, mask1=None):
    """(rowstat [Z,L,2], colstat [Z,S,2]) = (max, sum-exp) of the similarity matrix along each axis."""
    lib = _lib.load()
    Z, L, C = f0.shape
    S = f1.shape[1]
    ws = _ws(lib.far_dual_softmax_workspace_bytes(Z, L, S), f0.device)
    rowstat = torch.empty(Z, L, 2, dtype=torch.float32, device=f0.device)
    colstat = torch.empty(Z, S, 2, dtype=torch.float32, device=f0.device)
    rc = lib.far_dual_softmax_stats_f32(_p(f0, torch.float32), _p(f1, torch.float32), Z, L, S, C,
                                        feat_div, sim_div, sim_mul, _p(mask0, torch.uint8), _p(mask1, torch.uint8),
                                        _p(rowstat), _p(colstat), _p(ws), _stream())
    _lib.check(rc, 'far_dual_softmax_stats_f32')
    return rowstat, colstat


def coarse_match(f0, f1, temperature, thr, border, hw0, hw1, cell_scale, mask0=None, mask1=None,
                 valid_hw=None, scale0=None, scale1=None, want_conf=False, bf16=False, variant=None, overlap=None):
    """K1.  Returns dict(b_ids, i_ids, j_ids, mconf, mkpts0_c, mkpts1_c, counts, conf_matrix|None).
    variant: 'f32' exact-f32 MFMA (default), 'f16s' split-fp16 operands (fp32-grade), 'bf16' bf16 operands.

    One host synchronisation (reading M) is inherent: the reference's outputs have data-dependent shape
    (torch.where, coarse_matching.py:193).  overlap: a callable that enqueues work which does not depend on the matches; it runs
    between the (asynchronous) copy of the counts and the wait for it, so the GPU has that work to do while the host reads M and
    prepares the launches that depend on it.
    """
    lib = _lib.load()
    Z, L, C = f0.shape
    S = f1.shape[1]
    dev = f0.device
    variant = variant or ('bf16' if bf16 else 'f32')
    fn = {'f32': lib.far_coarse_match_f32, 'bf16': lib.far_coarse_match_bf16, 'f16s': lib.far_coarse_match_f16s}[variant]
    ws = _ws({'f32': lambda: lib.far_dual_softmax_workspace_bytes(Z, L, S),
              'bf16': lambda: lib.far_coarse_match_bf16_workspace_bytes(Z, L, S, C),
              'f16s': lambda: lib.far_coarse_match_f16s_workspace_bytes(Z, L, S, C)}[variant](), dev)
    cap = Z * L
    b_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    i_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    j_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    mconf = torch.empty(cap, dtype=torch.float32, device=dev)
    mk0 = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    mk1 = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    counts = torch.empty(Z + 1, dtype=torch.int32, device=dev)
    conf = torch.empty(Z, L, S, dtype=torch.float32, device=dev) if want_conf else None
    rc = fn(
        _p(f0, torch.float32), _p(f1, torch.float32), Z, L, S, C, float(temperature), float(thr), int(border),
        int(hw0[0]), int(hw0[1]), int(hw1[0]), int(hw1[1]), float(cell_scale),
        _p(mask0, torch.uint8), _p(mask1, torch.uint8), _p(valid_hw, torch.int32),
        _p(scale0, torch.float32), _p(scale1, torch.float32), _p(conf),
        _p(b_ids), _p(i_ids), _p(j_ids), _p(mconf), _p(mk0), _p(mk1),
        _p(counts), ctypes.c_void_p(counts.data_ptr() + 4 * Z), _p(ws),
        *([_p(overflow_flag(dev))] if variant == 'f16s' else []), _stream())
    _lib.check(rc, 'far_coarse_match_' + variant)
    if overlap is None:
        counts_h = counts.cpu()
    else:
        counts_h = torch.empty(Z + 1, dtype=torch.int32, pin_memory=True)
        counts_h.copy_(counts, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        overlap()
        done.synchronize()
    M = int(counts_h[Z])
    return {
        'b_ids': b_ids[:M], 'i_ids': i_ids[:M], 'j_ids': j_ids[:M], 'mconf': mconf[:M],
        'mkpts0_c': mk0[:M], 'mkpts1_c': mk1[:M], 'counts': counts_h[:Z], 'conf_matrix': conf,
    }


def conf_matrix(f0, f1, temperature, mask0=None, mask1=None, out=None):
    """K1, materialising mode: data['conf_matrix'] (Z, L, S) alone (coarse_matching.py:108-118) at HBM write speed
    (far_conf_matrix_f16s: fp32-grade statistics, plain-fp16 scores, exact recomputation of every entry above 2^-12).
    Falls back to the fused split-precision matcher's writer if the exact-entry list overflowed; reading that flag is one
    host synchronisation per call.  With `out=` the result is always in `out` (also after the fallback).
    Returns (conf, listed) with listed = number of entries that were recomputed exactly."""
    lib = _lib.load()
    Z, L, C = f0.shape
    S = f1.shape[1]
    dev = f0.device
    ws = _ws(lib.far_coarse_match_f16s_workspace_bytes(Z, L, S, C), dev)
    conf = torch.empty(Z, L, S, dtype=torch.float32, device=dev) if out is None else out
    info = torch.zeros(2, dtype=torch.int32, device=dev)
    rc = lib.far_conf_matrix_f16s(_p(f0, torch.float32), _p(f1, torch.float32), Z, L, S, C, float(temperature),
                                  _p(mask0, torch.uint8), _p(mask1, torch.uint8), 3, _p(conf, torch.float32), _p(info), _p(ws),
                                  _p(overflow_flag(dev)), _stream())
    _lib.check(rc, 'far_conf_matrix_f16s')
    listed, dropped = (int(v) for v in info.cpu())            # one blocking host read per call (the overflow flag)
    if dropped > 0:           # pathological input (a column with more than 8 non-tiny entries): the exact writer
        hw = (1, L), (1, S)
        exact = coarse_match(f0, f1, temperature, 2.0, 0, hw[0], hw[1], 1.0, mask0, mask1, want_conf=True,
                             variant='f16s')['conf_matrix']
        if out is None:
            return exact, listed
        out.copy_(exact)      # the caller's buffer must hold the result it asked for, not the partially exact one
        return _written(out), listed
    return (conf if out is None else _written(conf)), listed


class _CoarsePosConf(torch.autograd.Function):
    """conf_matrix[b, i, j] at M given positions, differentiable w.r.t. both coarse feature maps, without the dense
    matrix (far_coarse_pos_conf_f16s / far_coarse_pos_conf_bwd_f16)."""

    @staticmethod
    def forward(ctx, f0, f1, pb, pi, pj, temperature):
        lib = _lib.load()
        Z, L, C = f0.shape
        S = f1.shape[1]
        f0c, f1c = f0.detach().float().contiguous(), f1.detach().float().contiguous()
        pb, pi, pj = (t.to(torch.int64).contiguous() for t in (pb, pi, pj))
        M = int(pb.numel())
        ws = _ws(lib.far_coarse_train_workspace_bytes(Z, L, S, C), f0.device)
        p = torch.empty(M, dtype=torch.float32, device=f0.device)
        rc = lib.far_coarse_pos_conf_f16s(_p(f0c, torch.float32), _p(f1c, torch.float32), Z, L, S, C, float(temperature),
                                          _p(pb), _p(pi), _p(pj), M, _p(p), _p(ws), _p(overflow_flag(f0.device)), _stream())
        _lib.check(rc, 'far_coarse_pos_conf_f16s')
        ctx.save_for_backward(f0c, f1c, pb, pi, pj, p, ws)
        ctx.temperature = float(temperature)
        return p

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        f0c, f1c, pb, pi, pj, p, ws = ctx.saved_tensors
        Z, L, C = f0c.shape
        S = f1c.shape[1]
        w = (g.float() * p).contiguous()                    # dL/dp * p: bounded for the focal loss even where p -> 0
        df0, df1 = torch.empty_like(f0c), torch.empty_like(f1c)
        rc = lib.far_coarse_pos_conf_bwd_f16(_p(f0c), _p(f1c), Z, L, S, C, ctx.temperature, _p(pb), _p(pi), _p(pj),
                                             int(pb.numel()), _p(w, torch.float32), _p(df0), _p(df1), _p(ws), _stream())
        _lib.check(rc, 'far_coarse_pos_conf_bwd_f16')
        return df0, df1, None, None, None, None


def coarse_pos_conf(f0, f1, pb, pi, pj, temperature):
    """K1, training: conf_matrix[pb, pi, pj] (M,) fp32 with a HIP backward to both feature maps; C must be 256."""
    if not f0.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    return _CoarsePosConf.apply(f0, f1, pb, pi, pj, temperature)


def emm_bilinear(q, k, v, pos, scale, exact_f32=False, plain16=False):
    """K2.  q, k, v: (Z, N, 64) fp32; pos: (N, 6).  Returns F (Z, 70, 70) = v~^T (P v~), v~ = [v | pos],
    P = softmax(s, -1) * softmax(s, -2), s = (q k^T) * scale   (transformer.py:275-292).
    Default: split-fp16 operands on the f16 matrix cores (fp32-grade); exact_f32: the exact-f32 MFMA kernels;
    plain16: plain fp16 operands (far_emm_pv_f16, the 16-bit-operand class)."""
    lib = _lib.load()
    Z, N, D = q.shape
    T = torch.empty(Z, N, 70, dtype=torch.float32, device=q.device)
    if exact_f32:
        rowstat, colstat = dual_softmax_stats(q, k, 1.0, 1.0, scale)
        rc = lib.far_emm_pv_f32(_p(q, torch.float32), _p(k, torch.float32), _p(v, torch.float32), _p(pos, torch.float32),
                                Z, N, D, float(scale), _p(rowstat), _p(colstat), _p(T), _stream())
        _lib.check(rc, 'far_emm_pv_f32')
    else:
        ws = _ws(lib.far_emm_pv_f16s_workspace_bytes(Z, N), q.device)
        name = 'far_emm_pv_f16' if plain16 else 'far_emm_pv_f16s'
        rc = getattr(lib, name)(_p(q, torch.float32), _p(k, torch.float32), _p(v, torch.float32), _p(pos, torch.float32),
                                Z, N, D, float(scale), 1, 0, N * D, 0, _p(ws), _p(T), _p(overflow_flag(T.device)), _stream())
        _lib.check(rc, name)
    return emm_contract(_p(v, torch.float32), 1, 0, N * D, pos, T), T


def _emm_pv(q, k, v, pos, scale, want_stats=False):
    """far_emm_pv_f16s on contiguous (Z, N, 64) operands -> T = P [v | pos] (Z, N, 70) (+ the softmax statistics)."""
    lib = _lib.load()
    Z, N, D = q.shape
    T = torch.empty(Z, N, 70, dtype=torch.float32, device=q.device)
    ws = _ws(lib.far_emm_pv_f16s_workspace_bytes(Z, N), q.device)
    rc = lib.far_emm_pv_f16s(_p(q, torch.float32), _p(k, torch.float32), _p(v, torch.float32), _p(pos, torch.float32),
                             Z, N, D, float(scale), 1, 0, N * D, 0, _p(ws), _p(T), _p(overflow_flag(T.device)), _stream())
    _lib.check(rc, 'far_emm_pv_f16s')
    if not want_stats:
        return T
    rs = torch.empty(Z, N, 2, dtype=torch.float32, device=q.device)
    cs = torch.empty(Z, N, 2, dtype=torch.float32, device=q.device)
    rc = lib.far_emm_pv_f16s_copy_stats(_p(ws), Z, N, _p(rs), _p(cs), _stream())
    _lib.check(rc, 'far_emm_pv_f16s_copy_stats')
    return T, rs, cs


class _EmmBilinearFn(torch.autograd.Function):
    """F = vt^T P vt of the EMM head (K2) with its HIP backward: dq, dk from far_emm_bwd_f16 (recomputed score / dP
    tiles on the f16 matrix cores), dv from two (N x 70)(70 x 70) products; no (Z, N, N) tensor in either direction."""

    @staticmethod
    def forward(ctx, q, k, v, pos, scale):
        qc, kc, vc = (t.detach().float().contiguous() for t in (q, k, v))
        posc = pos.detach().float().contiguous()
        T, rs, cs = _emm_pv(qc, kc, vc, posc, scale, want_stats=True)
        vt = torch.cat([vc, posc.unsqueeze(0).expand(vc.shape[0], -1, -1)], dim=2)           # (Z, N, 70)
        ctx.save_for_backward(qc, kc, vc, posc, T, rs, cs)
        ctx.scale = float(scale)
        return torch.bmm(vt.transpose(1, 2), T)

    @staticmethod
    def backward(ctx, dF):
        lib = _lib.load()
        qc, kc, vc, posc, T, rs, cs = ctx.saved_tensors
        Z, N, D = qc.shape
        dF = dF.float().contiguous()
        vt = torch.cat([vc, posc.unsqueeze(0).expand(Z, -1, -1)], dim=2).contiguous()
        # T' = P^T vt: the forward kernel with the roles of q and k exchanged (the dual softmax is symmetric under it)
        Tp = _emm_pv(kc, qc, vc, posc, ctx.scale)
        A = torch.bmm(vt, dF)                                                                  # vt dF
        Bm = torch.bmm(vt, dF.transpose(1, 2))                                                 # vt dF^T
        dvt = torch.bmm(T, dF.transpose(1, 2)) + torch.bmm(Tp, dF)
        u = (A * T).sum(-1)
        vw = (Bm * Tp).sum(-1)
        # A common power-of-two scale alpha on (A, u, v) -- ds is linear in them -- places the kernel's fp16 quantities:
        # ds = 2 P dP - R u - C v is bounded by Rmax (2 |dP|max + |u|max) + Cmax |v|max with Rmax = 1 / min rowsum,
        # Cmax = 1 / min colsum and |dP| <= max |A_a| max |v~_b|; alpha brings that bound to 2^14 (for diffuse attention
        # the typical entry sits N times lower: still a normal fp16 number -- without this ds underflowed at N = 4800),
        # capped so that the operand A * 2^4 stays below 2^15.  All of it device-side scalars: no host round trip.
        tiny = 1e-30
        bound = (1.0 / rs[..., 1].amin()) * (2.0 * A.norm(dim=-1).amax() * vt.norm(dim=-1).amax() + u.abs().amax()) \
            + (1.0 / cs[..., 1].amin()) * vw.abs().amax()
        alpha = torch.minimum(2.0 ** 11 / A.abs().amax().clamp_min(tiny), 2.0 ** 14 / bound.clamp_min(tiny))
        alpha = torch.exp2(torch.floor(torch.log2(alpha)))
        u = (u * alpha).contiguous()
        vw = (vw * alpha).contiguous()
        A = (A * alpha).contiguous()
        dq, dk = torch.empty_like(qc), torch.empty_like(kc)
        ws = _ws(lib.far_emm_bwd_workspace_bytes(Z, N), qc.device)
        rc = lib.far_emm_bwd_f16(_p(qc), _p(kc), _p(vt, torch.float32), _p(A, torch.float32), _p(u, torch.float32),
                                 _p(vw, torch.float32), _p(rs), _p(cs), Z, N, ctx.scale, _p(dq), _p(dk), _p(ws), _stream())
        _lib.check(rc, 'far_emm_bwd_f16')
        inv = 1.0 / alpha
        return dq * inv, dk * inv, dvt[:, :, :D].contiguous(), None, None


def emm_bilinear_train(q, k, v, pos, scale):
    """K2 with gradients: q, k, v (Z, N, 64), pos (N, 6) -> F (Z, 70, 70)."""
    if not q.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    return _EmmBilinearFn.apply(q, k, v, pos, scale)


def emm_bilinear_planes(qkv, pos, scale, B, plain16=False):
    """K2 on the output of the head's fused q | k | v projection: qkv (12, 2B, N, 64) = (tensor t, head) planes of
    [image, pair][N][64] (ops.linear_f16s(..., out_planes=12)).  Problem z = (direction, pair, head); direction d pairs
    the queries of image 1 - d with the keys / values of image d (transformer.py:275-276, 291-292).
    plain16: far_emm_pv_f16 (plain fp16 operands) instead of the split-fp16 far_emm_pv_f16s.
    Returns F (2 B h, 70, 70), T."""
    lib = _lib.load()
    P12, P, N, D = qkv.shape
    h = P12 // 3
    Z = P * h
    T = torch.empty(Z, N, 70, dtype=torch.float32, device=qkv.device)
    ws = _ws(lib.far_emm_pv_f16s_workspace_bytes(Z, N), qkv.device)
    base = qkv.data_ptr()
    plane = P * N * D * 4
    name = 'far_emm_pv_f16' if plain16 else 'far_emm_pv_f16s'
    rc = getattr(lib, name)(ctypes.c_void_p(base), ctypes.c_void_p(base + h * plane), ctypes.c_void_p(base + 2 * h * plane),
                            _p(pos, torch.float32), Z, N, D, float(scale), h, P * N * D, N * D, B, _p(ws), _p(T), _p(overflow_flag(T.device)), _stream())
    _lib.check(rc, name)
    # F = [v | pos]^T T (transformer.py:291-295) straight from the v planes: no (Z, N, 70) concatenation, no vendor bmm
    return emm_contract(ctypes.c_void_p(base + 2 * h * plane), h, P * N * D, N * D, pos, T), T


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


def train_pack(cache, name, weight, bias=None, split=True):
    """The K9 image of a Linear layer's weight for the training forward (re-packed when the weight's version changes)."""
    return cache.get((name, split), [weight] + ([bias] if bias is not None else []), lambda: PackedConv(weight, None, bias, split=split),
                     refresh=(lambda pc: pc.refresh(weight)) if bias is None else None)


def train_pack_t(cache, name, weight, bias=None, split=True):
    """The transposed image (dgrad): the same tensor read through strides, with the forward image's scale (same maximum)."""
    fwd = train_pack(cache, name, weight, bias, split)             # first: the transposed image borrows its (refreshed) scale
    pt = cache.get((name, 'T', split), [weight], lambda: PackedConv(weight, split=split, dgrad=True, pack_scale=fwd.pack_scale),
                   refresh=lambda pc: pc.refresh(weight))
    return pt.follow_scale(fwd, weight)


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


def solve_pose_batch(kpts0, kpts1, offsets_host, K0, K1, inl_th, many_thr, priorRT=None, pcl=None,
                     prior_lambda=0.3, H=2048, seed=0, samples=None, debug=False, minimal=8):
    """K4.  kpts0/kpts1: (Mtot, 2) fp32 GPU; offsets_host: python list / CPU int tensor of B+1 offsets;
    K0/K1: (B, 3, 3) float64 GPU; inl_th: (B,) float64 GPU; priorRT: (B, 3, 4) fp32 GPU or None.
    minimal: 8 = normalized 8-point hypotheses (pairs with 5..7 correspondences: five-point), 5 = Nister's five-point solver
    for every pair; H = models verified per pair (a five-point sample yields up to ten: H // 10 samples).
    Returns a dict of GPU tensors (R, t, E float64; mask uint8; status/num_after/tight/ultra/n_cheir/best int32)."""
    lib = _lib.load()
    dev = K0.device
    offs = torch.as_tensor(offsets_host, dtype=torch.int32)
    B = offs.numel() - 1
    Mtot = int(offs[-1])
    Mmax = int((offs[1:] - offs[:-1]).max()) if B > 0 else 0
    offs_d = offs.pin_memory().to(dev, non_blocking=True)          # pageable memory would make this upload a stream synchronisation
    P = 0 if pcl is None else int(pcl.shape[0])
    ws = _ws(lib.far_solver_workspace_bytes(B, Mtot, H, P), dev)
    f64, i32 = torch.float64, torch.int32
    out = {
        'R': torch.empty(B, 3, 3, dtype=f64, device=dev), 't': torch.empty(B, 3, dtype=f64, device=dev),
        'E': torch.empty(B, 3, 3, dtype=f64, device=dev), 'mask': torch.empty(Mtot, dtype=torch.uint8, device=dev),
        'status': torch.empty(B, dtype=i32, device=dev), 'num_after': torch.empty(B, dtype=i32, device=dev),
        'tight': torch.empty(B, dtype=i32, device=dev), 'ultra': torch.empty(B, dtype=i32, device=dev),
        'n_cheir': torch.empty(B, dtype=i32, device=dev), 'best': torch.empty(B, dtype=i32, device=dev),
    }
    dbg = {}
    if debug:
        dbg = {'F_all': torch.empty(B, H, 3, 3, dtype=f64, device=dev), 'count_all': torch.empty(B, H, dtype=i32, device=dev),
               'score_all': torch.empty(B, H, dtype=f64, device=dev),
               'samples': torch.full((B, H, 8) if minimal == 8 else (B, max(H // 10, 1), 5), -1, dtype=i32, device=dev)}
    rc = lib.far_solver_f64(
        _p(kpts0, torch.float32) if Mtot else ctypes.c_void_p(0), _p(kpts1, torch.float32) if Mtot else ctypes.c_void_p(0),
        _p(offs_d), B, Mtot, Mmax, _p(K0.contiguous(), f64), _p(K1.contiguous(), f64), _p(inl_th, f64), int(bool(many_thr)),
        _p(priorRT, torch.float32), _p(pcl, torch.float32), P, float(prior_lambda), int(H), int(minimal), int(seed) & 0xffffffff,
        _p(samples, torch.int32),
        _p(out['R']), _p(out['t']), _p(out['E']), _p(out['mask']), _p(out['status']), _p(out['num_after']),
        _p(out['tight']), _p(out['ultra']), _p(out['n_cheir']), _p(out['best']),
        _p(dbg.get('F_all')), _p(dbg.get('count_all')), _p(dbg.get('score_all')), _p(dbg.get('samples')),
        _p(ws), _stream())
    _lib.check(rc, 'far_solver_f64')
    out.update(dbg)
    out['offsets'] = offs_d
    return out


def prior_from_pose(pose, mean, std):
    """K11c.  pose (B, 9) fp32 GPU (normalised [t | 6D rotation]), mean / std (9,) fp32 GPU -> (B, 3, 4) fp32 [R | t]: the head's pose as the
    next solver round's prior (loftr.py:186-192) in one launch."""
    lib = _lib.load()
    pose = pose.detach().float().contiguous()
    B = pose.shape[0]
    out = torch.empty(B, 3, 4, dtype=torch.float32, device=pose.device)
    rc = lib.far_prior_from_pose_f32(_p(pose, torch.float32), _p(mean.contiguous(), torch.float32), _p(std.contiguous(), torch.float32), B,
                                     _p(out), _stream())
    _lib.check(rc, 'far_prior_from_pose_f32')
    return out


def pose_pack(sol, offsets_dev):
    """K11a.  The solver's result dict -> the data-dict tensors of spvs_RT (supervision.py:218-233) in one launch:
    rt (B, 3, 4) and E (B, 3, 3) float64 with the identity fallback, before (B,) int64, after / tight / ultra (B,) int32
    (zero for pairs with fewer than 5 correspondences).  offsets_dev: the solver's (B + 1,) int32 offsets on the GPU."""
    lib = _lib.load()
    B = sol['R'].shape[0]
    dev = sol['R'].device
    i32 = torch.int32
    rt = torch.empty(B, 3, 4, dtype=torch.float64, device=dev)
    E = torch.empty(B, 3, 3, dtype=torch.float64, device=dev)
    before = torch.empty(B, dtype=torch.int64, device=dev)
    after, tight, ultra = (torch.empty(B, dtype=i32, device=dev) for _ in range(3))
    rc = lib.far_pose_pack_f64(_p(sol['R'], torch.float64), _p(sol['t'], torch.float64), _p(sol['E'], torch.float64),
                               _p(sol['status'], i32), _p(sol['num_after'], i32), _p(sol['tight'], i32), _p(sol['ultra'], i32),
                               _p(offsets_dev, i32), B, _p(rt), _p(E), _p(before), _p(after), _p(tight), _p(ultra), _stream())
    _lib.check(rc, 'far_pose_pack_f64')
    return rt, E, before, after, tight, ultra


def pose_features(rt, counts=()):
    """K11b.  preprocess_helper's arithmetic (loftr.py:137-171): rt (B, 3, 4) float64 GPU -> (preds, inv_preds), each
    (B, 9 + len(counts)) fp32: the pose / its inverse as normalised [t, R rows 0-1], then count / 500 per count vector
    (each (B,) int32 or int64 on the GPU, at most four)."""
    lib = _lib.load()
    B = rt.shape[0]
    if len(counts) > 4:
        raise _lib.FarHipError('pose_features: at most four count vectors')
    args = []
    for c in counts:
        if c.dtype not in (torch.int32, torch.int64) or c.numel() != B:
            raise _lib.FarHipError('pose_features: counts must be (B,) int32 / int64 tensors')
        args += [_p(c.contiguous()), c.element_size()]
    args += [ctypes.c_void_p(0), 0] * (4 - len(counts))
    width = 9 + len(counts)
    preds = torch.empty(B, width, dtype=torch.float32, device=rt.device)
    inv = torch.empty(B, width, dtype=torch.float32, device=rt.device)
    rc = lib.far_pose_features_f32(_p(rt.contiguous(), torch.float64), B, *args, _p(preds), _p(inv), _stream())
    _lib.check(rc, 'far_pose_features_f32')
    return preds, inv


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


def _layout(t):
    """0 = contiguous NCHW, 1 = channels_last memory; anything else is re-laid out as NCHW by the caller."""
    if t.is_contiguous():
        return 0
    if t.is_contiguous(memory_format=torch.channels_last):
        return 1
    return -1


def _same_layout(ts):
    """Bring 4-D fp32 GPU tensors to one memory layout (the first tensor's, NCHW if it has neither)."""
    for t in ts:
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
            raise _lib.FarHipError('backbone epilogues need fp32 GPU (N, C, H, W) tensors')
    lay = _layout(ts[0])
    if lay < 0:
        lay = 0
    fmt = torch.channels_last if lay == 1 else torch.contiguous_format
    return lay, [t if _layout(t) == lay else t.contiguous(memory_format=fmt) for t in ts]


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


def _raw(t):
    """Device pointer of a tensor whose memory layout the caller has established (channels_last 4-D tensors are not `contiguous`)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


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


_ACT = {'none': 0, 'relu': 1, 'leaky': 2}
_CONV_ACT_EXP = 4          # default exponent of the activation scale 2^e applied before the fp16 split (conv_igemm_f16s.hip);
                           # the packed `scale` vectors fold 2^-4, the kernel corrects for the exponent actually used


# ---------------------------------------------------------------------------------------------------------------------
# Activation range of the split-precision kernels (K9, K13, K14).  An fp32 activation a is split hi = fp16(a 2^e),
# lo = fp16(a 2^e - hi): |a| <= 65504 / 2^e survives, beyond it hi = inf.  The reference's fp32 convolutions / Linear layers
# have no such limit (resnet_fpn.py:101-119, transformer.py:44-67), so the limit must never bite silently:
#   * every launch tests its accumulators and ORs a per-device flag (no cost unless it fires);
#   * `check_activation_range` reads the flag (one host read) and raises ActivationOverflow;
#   * far_amd.loftr.LoFTR catches it, lowers the exponent e (thread-local, `activation_exponent`) by 4 -- 16x the range,
#     16x coarser absolute resolution of values below 2^-14 2^-e -- switches the fused fine-level layers (whose exponent
#     is fixed) to their K9 + K5 form, and re-runs the forward.  e = 4 covers |a| <= 4094; the floor e = -24 covers 1e12.
# ---------------------------------------------------------------------------------------------------------------------
class ActivationOverflow(_lib.FarHipError):
    pass


_ACT_STATE = threading.local()
_OVERFLOW_FLAGS = {}
ACT_EXP_MIN = -24


def activation_exponent_value():
    return getattr(_ACT_STATE, 'exp', _CONV_ACT_EXP)


class activation_exponent:
    """Context manager: K9 launches of this thread split their activations around 2^exp."""

    def __init__(self, exp):
        if not (ACT_EXP_MIN <= int(exp) <= 8):
            raise ValueError(f'activation exponent must be in [{ACT_EXP_MIN}, 8]')
        self.exp = int(exp)

    def __enter__(self):
        self.prev = activation_exponent_value()
        _ACT_STATE.exp = self.exp
        return self

    def __exit__(self, *a):
        _ACT_STATE.exp = self.prev


def overflow_flag(device):
    """The per-device int32 flag every K9 / K13 / K14 launch ORs into."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    f = _OVERFLOW_FLAGS.get(key)
    if f is None:
        f = _OVERFLOW_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=torch.device('cuda', key))
    return f


def activation_overflowed(device, reset=True):
    """True if a launch since the last reset saw a non-finite accumulator.  One blocking host read."""
    f = overflow_flag(device)
    hit = bool(f.item())
    if hit and reset:
        f.zero_()
    return hit


def check_activation_range(device, what='far_amd'):
    if activation_overflowed(device):
        raise ActivationOverflow(
            f'{what}: an activation left the range of the split-fp16 operands (|a| > {65504.0 / 2.0 ** activation_exponent_value():.4g} '
            f'at activation exponent {activation_exponent_value()}): the outputs of this call contain inf / NaN.  '
            f'Re-run under ops.activation_exponent(e) with a lower e (far_amd.loftr.LoFTR does this by itself).')


class PackedConv:
    """Weights of one convolution / linear layer in K9's packed split-fp16 image, plus the folded epilogue vectors."""

    def __init__(self, weight, scale=None, shift=None, split=True, stride=1, dgrad=False, pack_scale=None):
        """weight: (Cout, Cin, k, k) or (Cout, Cin).  dgrad=True packs the image of the layer's input-gradient convolution --
        channels exchanged, taps reversed (a Linear layer: the transposed weight) -- read from the SAME tensor through strides.
        pack_scale: the two device floats of another image of the same weight (its maximum is the same): skips the reduction."""
        lib = _lib.load()
        w = weight.detach()
        if w.dim() == 2:
            w = w[:, :, None, None]
        Cout, Cin, kh, kw = w.shape
        if kh != kw or kh not in (1, 3):
            raise _lib.FarHipError(f'K9 supports 1x1 and 3x3 kernels, got {kh}x{kw}')
        w = w.contiguous().float()
        if stride not in (1, 2) or (stride == 2 and kh != 3):
            raise _lib.FarHipError('K9 supports stride 1, and stride 2 for 3x3 kernels')
        T = kh * kh
        if dgrad:
            view = (T, Cin * T, -1 if T > 1 else 0, T - 1)          # (s_co, s_ci, s_tap, offset of tap 0) of the dgrad image
            Cin, Cout = Cout, Cin
        else:
            view = (Cin * T, T, 1, 0)
        self.Cin, self.Cout, self.ksize, self.split, self.stride = Cin, Cout, kh, bool(split), stride
        nbytes = lib.far_conv_packed_bytes(Cin, Cout, kh, stride, int(self.split))
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        # the power-of-two weight scale 2^w_exp (max |w| 2^w_exp in [2^13, 2^14)) is chosen on the device: no host read of the
        # weights, so re-packing after every optimizer step costs two small launches and no synchronisation
        self._own_scale = pack_scale is None
        self.pack_scale = torch.empty(2, dtype=torch.float32, device=w.device) if pack_scale is None else pack_scale   # { 2^w_exp, 2^-(w_exp + 4) }
        self._view, self._wshape = view, tuple(w.shape)
        self._base = None if scale is None else scale.detach().float().contiguous()
        self.scale = torch.empty(Cout, dtype=torch.float32, device=w.device)      # base scale x 2^-(w_exp + 4), written by the pack kernel
        self.shift = None if shift is None else shift.detach().float().contiguous()
        self._wino, self._wino_stale = None, False
        self._pack(w)

    def _pack(self, w):
        lib = _lib.load()
        if self._own_scale:
            _lib.check(lib.far_weight_scale_f32(_p(w, torch.float32), w.numel(), _p(self.pack_scale), _stream()), 'far_weight_scale_f32')
        v = self._view
        rc = lib.far_conv_pack_view_scaled_f32(ctypes.c_void_p(w.data_ptr() + 4 * v[3]), v[0], v[1], v[2], self.Cin, self.Cout, self.ksize,
                                               self.stride, int(self.split), _p(self.pack_scale), _p(self.packed),
                                               _p(self._base) if self._base is not None else None, _p(self.scale), _stream())
        _lib.check(rc, 'far_conv_pack_view_scaled_f32')
        self._w = w                                               # keeps the (possibly temporary) contiguous weight alive until the pack ran

    def follow_scale(self, owner, weight):
        """For an image that borrows another image's pack_scale (dgrad / transposed images): when the owner was REBUILT rather than
        refreshed (a biased layer, a changed stamp) it holds a new scale tensor and this image would keep packing with the orphaned,
        never-updated one -- rebind to the owner's current tensor and re-pack.  Returns self."""
        if not self._own_scale and self.pack_scale.data_ptr() != owner.pack_scale.data_ptr():
            self.pack_scale = owner.pack_scale
            self.refresh(weight)
            PACK_TABLE.dirty = True                     # the device table holds the old scale pointer
        return self

    def refresh(self, weight):
        """Re-pack in place after the weight changed (an optimizer step): the same buffers, one launch (+ the scale reduction when
        this image owns it; an image that borrows another's pack_scale must be refreshed after that one).  Epilogue scale / shift
        vectors passed at construction are kept as they were."""
        w = weight.detach()
        if w.dim() == 2:
            w = w[:, :, None, None]
        if tuple(w.shape) != self._wshape or w.dtype != torch.float32 or not w.is_contiguous() or w.device != self.packed.device:
            raise _lib.FarHipError('PackedConv.refresh: the weight changed shape, dtype, layout or device')
        self._pack(w)
        self.invalidate_wino()
        return self

    def invalidate_wino(self):
        """The weight changed: the K17 image (if one was built) holds the old weights.  It is re-packed in place at its next use
        (wino()); every path that re-packs this image -- refresh() and the whole-model table (_PackTable.refresh_all) -- ends here."""
        if self._wino:
            self._wino_stale = True

    def wino(self):
        """The K17 image of the same layer (built at the first inference launch that can use it, from the weight this image was
        packed from); None for layers K17 does not serve (1x1, stride 2, plain-fp16 operands, dgrad images, channel counts not
        divisible by four)."""
        if self._wino is None:
            ok = (self.ksize == 3 and self.stride == 1 and self.split and self._view[3] == 0 and self.Cin % 4 == 0 and self.Cout % 4 == 0
                  and self._w.dim() == 4)
            self._wino = PackedWino(self._w, self._base, self.shift) if ok else False
            self._wino_stale = False
        elif self._wino and self._wino_stale:
            self._wino.refresh(self._w)                           # self._w shares the parameter's storage: the current weights
            self._wino_stale = False
        return self._wino or None


class PackedWino:
    """Weights of one stride-1 3x3 convolution in K17's Winograd image (U = G g G^T, split fp16 planes), plus the folded epilogue
    vectors; the same constructor meaning as PackedConv (scale / shift: the inference BatchNorm as a per-channel affine map)."""

    ksize, stride, split = 3, 1, True

    def __init__(self, weight, scale=None, shift=None):
        lib = _lib.load()
        w = weight.detach()
        if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
            raise _lib.FarHipError(f'K17 is a 3x3 kernel, got a weight of shape {tuple(w.shape)}')
        w = w.contiguous().float()
        self.Cout, self.Cin = int(w.shape[0]), int(w.shape[1])
        if self.Cin % 4:
            raise _lib.FarHipError('K17 needs Cin % 4 == 0')
        self.packed = torch.empty(lib.far_wino_packed_bytes(self.Cin, self.Cout), dtype=torch.uint8, device=w.device)
        self.pack_scale = torch.empty(2, dtype=torch.float32, device=w.device)       # { 2^w_exp, 2^-(w_exp + 4) }
        self._base = None if scale is None else scale.detach().float().contiguous()
        self.scale = torch.empty(self.Cout, dtype=torch.float32, device=w.device)
        self.shift = None if shift is None else shift.detach().float().contiguous()
        self._wshape = tuple(w.shape)
        self.refresh(w)

    def refresh(self, weight):
        """(Re-)packs the image from `weight` into the same buffers: two launches, no allocation."""
        lib = _lib.load()
        w = weight.detach()
        if tuple(w.shape) != self._wshape or w.dtype != torch.float32 or not w.is_contiguous() or w.device != self.packed.device:
            raise _lib.FarHipError('PackedWino.refresh: the weight changed shape, dtype, layout or device')
        _lib.check(lib.far_weight_scale_f32(_p(w, torch.float32), w.numel(), _p(self.pack_scale), _stream()), 'far_weight_scale_f32')
        rc = lib.far_wino_pack_view_scaled_f32(_p(w), 9 * self.Cin, 9, 1, self.Cin, self.Cout, _p(self.pack_scale), _p(self.packed),
                                               _p(self._base) if self._base is not None else None, _p(self.scale), _stream())
        _lib.check(rc, 'far_wino_pack_view_scaled_f32')
        self._w = w
        return self


WINO_MIN_ACT_EXP = 0        # K17 splits its operands unscaled (|a| <= 16376): used while the activation exponent is >= 0
WINO_MIN_PIXELS = 1024      # per image; below, a 16x16-output workgroup tile is mostly padding
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


class _PackTable:
    """Every refreshable weight image of the process (training: PackedConv objects whose cache entry depends on the weight
    alone), re-packed together after an optimizer step: far_pack_table_run = two launches for all of them instead of two per
    image.  Entries are weak: an image lives as long as the PackCache of its module does."""

    def __init__(self):
        self.entries = []          # (weakref(cache), key, weakref(weight), weakref(pc))
        self.table = None          # (device table tensor, n, [(weakref(cache), key, weakref(weight), weakref(pc))]): weak, like entries
        self.dirty = True
        self._skip = 0             # stale lookups still to come in the step for which the per-entry path was chosen

    def register(self, cache, key, weight, pc):
        import weakref
        self.entries.append((weakref.ref(cache), key, weakref.ref(weight), weakref.ref(pc)))
        self.dirty = True

    def _live(self):
        out = []
        for e in self.entries:
            cache, w, pc = e[0](), e[2](), e[3]()
            if cache is not None and w is not None and pc is not None and cache._store.get(e[1], (None, None))[1] is pc:
                out.append((cache, e[1], w, pc))
        return out

    def _build(self, live):
        import weakref
        lib = _lib.load()
        dev = live[0][3].packed.device
        live = [e for e in live if e[3].packed.device == dev and e[2].is_contiguous() and e[2].dtype == torch.float32]
        owner = {}
        for i, (_, _, _, pc) in enumerate(live):
            if pc._own_scale:
                owner[pc.pack_scale.data_ptr()] = i
        keep = [e for e in live if e[3].pack_scale.data_ptr() in owner]
        owner = {pc.pack_scale.data_ptr(): i for i, (_, _, _, pc) in enumerate(keep) if pc._own_scale}
        keep = [e for e in keep if e[3].pack_scale.data_ptr() in owner]          # (a borrower whose owner dropped out goes too)
        n = len(keep)
        if n == 0 or n > 4096:
            return None
        items = (_lib.PackItem * n)()
        for i, (_, _, w, pc) in enumerate(keep):
            v, it = pc._view, items[i]
            it.w, it.s_co, it.s_ci, it.s_tap = w.data_ptr() + 4 * v[3], v[0], v[1], v[2]
            it.Cin, it.Cout, it.ksize, it.stride, it.split = pc.Cin, pc.Cout, pc.ksize, pc.stride, int(pc.split)
            it.scale_owner = owner[pc.pack_scale.data_ptr()]
            it.w_all, it.n_all = w.data_ptr(), w.numel()
            it.pack_scale, it.packed = pc.pack_scale.data_ptr(), pc.packed.data_ptr()
            it.base_scale = pc._base.data_ptr() if pc._base is not None else None
            it.scale_vec = pc.scale.data_ptr()
        table = torch.empty(int(lib.far_pack_table_bytes(n)), dtype=torch.uint8, device=dev)
        _lib.check(lib.far_pack_table_build(ctypes.cast(items, ctypes.c_void_p), n, _p(table), _stream()), 'far_pack_table_build')
        # only weak references are kept next to the device table (which holds raw pointers): the table must not pin the weights,
        # images and caches of a model that was deleted; a dead reference found later marks the table dirty
        return table, n, [(weakref.ref(c), k, weakref.ref(w), weakref.ref(pc)) for c, k, w, pc in keep]

    def refresh_all(self):
        """Re-packs every live image whose weight version changed, through the table when most of them did.  Returns True when
        the table ran (the caller's entry is then fresh)."""
        if self._skip > 0:                              # the rest of a step's stale lookups after the per-entry path was chosen
            self._skip -= 1
            return False
        if self.dirty:
            live = self._live()
            self.entries = [e for e in self.entries if e[0]() is not None and e[3]() is not None]
            self.table = self._build(live) if live else None
            self.dirty = False
        if self.table is None:
            return False
        table, n, refs = self.table
        keep, stamps, stale = [], [], 0
        for rc, key, rw, rpc in refs:
            cache, w, pc = rc(), rw(), rpc()
            if cache is None or w is None or pc is None:
                self.dirty = True                       # a model went away: the table's raw pointers are stale, rebuild next time
                self.table = None
                return False
            st = ((w.data_ptr(), tensor_version(w)),)
            hit = cache._store.get(key)
            if hit is None or hit[1] is not pc or hit[0][0][0] != st[0][0]:
                self.dirty = True                       # an entry was replaced or its weight moved: rebuild next time, per-entry now
                return False
            keep.append((cache, key, w, pc))
            stamps.append(st)
            stale += hit[0] != st
        if 2 * stale < n:
            # a few images only (fine-tuning a sub-module): per-entry refresh -- and no second walk over all n entries for each of
            # the other stale images of this step (they each come through here once)
            self._skip = max(stale - 1, 0)
            return False
        _lib.check(_lib.load().far_pack_table_run(_p(table), n, _stream()), 'far_pack_table_run')
        for (cache, key, w, pc), st in zip(keep, stamps):
            cache._store[key] = (st, pc)
            pc.invalidate_wino()                        # the table re-packs K9's images only: K17's follow lazily, in place
        return True


PACK_TABLE = _PackTable()
USE_PACK_TABLE = True       # False: every stale image re-packs itself (two launches each)


class PackCache:
    """K9 weight images keyed by name, rebuilt when any tensor they were derived from changes (in-place update,
    load_state_dict, optimizer step: data_ptr / _version stamp)."""

    def __init__(self):
        self._store = {}

    def get(self, key, tensors, build, refresh=None):
        """refresh(obj): optional in-place update of the stored object when only tensor versions changed (same storage): a
        training step re-packs every weight, and reusing the buffers saves the allocations and two launches per image."""
        stamp = tuple((t.data_ptr(), tensor_version(t)) for t in tensors)
        hit = self._store.get(key)
        if hit is None or hit[0] != stamp:
            same_storage = hit is not None and refresh is not None and tuple(p for p, _ in hit[0]) == tuple(p for p, _ in stamp)
            if same_storage and USE_PACK_TABLE and len(tensors) == 1 and isinstance(hit[1], PackedConv) and PACK_TABLE.refresh_all():
                hit = self._store[key]                   # the whole model's images were re-packed together
                if hit[0] == stamp:
                    return hit[1]
            new = refresh(hit[1]) if same_storage else build()
            hit = (stamp, new)
            self._store[key] = hit
            if refresh is not None and not same_storage and len(tensors) == 1 and isinstance(new, PackedConv):
                PACK_TABLE.register(self, key, tensors[0], new)
        return hit[1]


def grad_scale(x):
    """Two device floats { 2^e, 2^(4 - e) }, max|x| 2^e in [2^9, 2^10): the activation scale of a K9 launch whose input is a
    gradient (conv_nhwc(..., act_scale_dev=...)); no host synchronisation."""
    lib = _lib.load()
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    _lib.check(lib.far_grad_scale_f32(_p(x, torch.float32), x.numel(), _p(out), _stream()), 'far_grad_scale_f32')
    return out


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


_CVW_GRID = {}


def corr_volume_warp(vol0, vol1):
    """K12.  vol0, vol1 (B, 32, H, W) fp32 -> agg (B, 67, H, W) = cat[vol0, warped vol1, warped (u, v) grid, max score]:
    CorrelationVolumeWarping.forward of the Map-free 6DReg model (aggregator.py:44-115) without the (B, HW, HW) volume."""
    lib = _lib.load()
    B, D, H, W = vol0.shape
    if vol1.shape != vol0.shape:
        raise _lib.FarHipError('Feature volumes shape must match')
    key = (H, W, str(vol0.device))
    if key not in _CVW_GRID:
        u = torch.linspace(-1, 1, H, device=vol0.device)
        v = torch.linspace(-1, 1, W, device=vol0.device)
        uu, vv = torch.meshgrid(u, v, indexing='ij')
        _CVW_GRID[key] = torch.stack([uu, vv], 0).reshape(2, H * W).contiguous()
    agg = torch.empty(B, 2 * D + 3, H, W, dtype=torch.float32, device=vol0.device)
    ws = _ws(lib.far_corr_volume_warp_workspace_bytes(B, H * W), vol0.device)
    rc = lib.far_corr_volume_warp_f32(_p(vol0.float().contiguous(), torch.float32), _p(vol1.float().contiguous(), torch.float32),
                                      _p(_CVW_GRID[key]), B, D, H * W, _p(agg), _p(ws), _stream())
    _lib.check(rc, 'far_corr_volume_warp_f32')
    return agg


# ---------------------------------------------------------------------------------------------------------------------
# K15: row-independent exact-fp32 layers of the regression head (head_linear_f32.hip)
# ---------------------------------------------------------------------------------------------------------------------
_ROWS_ACT = {'none': 0, 'relu': 1, 'sigmoid': 2, 'gelu': 3}


class PackedRows:
    """An nn.Linear's weight [N][K] (optionally a column range of it) in far_rows_linear_f32's [K / 4][N][4] image."""

    def __init__(self, weight, bias=None, cols=None):
        lib = _lib.load()
        w = weight.detach().float()
        if cols is not None:
            w = w[:, cols[0]:cols[1]]
        w = w.contiguous()
        self.N, self.K = int(w.shape[0]), int(w.shape[1])
        self.packed = torch.empty(lib.far_rows_linear_packed_bytes(self.N, self.K), dtype=torch.uint8, device=w.device)
        _lib.check(lib.far_rows_linear_pack_f32(_p(w, torch.float32), self.N, self.K, _p(self.packed), _stream()), 'far_rows_linear_pack_f32')
        self.bias = None if bias is None else bias.detach().float().contiguous()


def rows_linear(x, pr, act='none', add=None):
    """K15.  x (B, K) fp32 (rows may be strided) -> act(x W^T + bias + add) (B, N): every row one fp32 fma chain in a fixed
    order -- bit-identical whatever B is."""
    lib = _lib.load()
    if not x.is_cuda:
        raise _lib.FarHipError('far_amd ops need tensors on the GPU (no CPU fallback exists)')
    if x.dim() != 2 or x.shape[1] != pr.K or x.dtype != torch.float32 or x.stride(1) != 1:
        raise _lib.FarHipError(f'rows_linear: x must be (B, {pr.K}) fp32 with unit column stride')
    if add is not None and (tuple(add.shape) != (x.shape[0], pr.N) or add.stride(1) != 1 or add.dtype != torch.float32):
        raise _lib.FarHipError('rows_linear: add must be (B, N) fp32 with unit column stride')
    B = int(x.shape[0])
    y = torch.empty(B, pr.N, dtype=torch.float32, device=x.device)
    if B == 0:
        return y
    ws = _ws(lib.far_rows_linear_workspace_bytes(B, pr.N, pr.K), x.device)
    rc = lib.far_rows_linear_f32(ctypes.c_void_p(x.data_ptr()), int(x.stride(0)), _p(pr.packed), _p(pr.bias, torch.float32),
                                 ctypes.c_void_p(add.data_ptr()) if add is not None else ctypes.c_void_p(0),
                                 int(add.stride(0)) if add is not None else 0, B, pr.K, pr.N, _ROWS_ACT[act], _p(y), pr.N, _p(ws),
                                 _stream())
    _lib.check(rc, 'far_rows_linear_f32')
    return y


def emm_contract(v_ptr, heads, head_stride, prob_stride, pos, T):
    """K15.  F (Z, 70, 70) = [v | pos]^T T per problem; v addressed as far_emm_pv_f16s addresses it."""
    lib = _lib.load()
    Z, N, _ = T.shape
    F = torch.empty(Z, 70, 70, dtype=torch.float32, device=T.device)
    ws = _ws(lib.far_emm_contract_workspace_bytes(Z), T.device)
    rc = lib.far_emm_contract_f32(v_ptr, int(heads), int(head_stride), int(prob_stride), _p(pos, torch.float32), _p(T, torch.float32),
                                  Z, N, _p(F), _p(ws), _stream())
    _lib.check(rc, 'far_emm_contract_f32')
    return F
