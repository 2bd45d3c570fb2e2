"""far_amd.ops.coarse: K1: statistics, the coarse matcher, conf_matrix, the sparse-position training form (one family of the torch-tensor front ends for the C ABI in include/far_hip.h; far_amd/ops/__init__.py
re-exports everything under the flat far_amd.ops namespace the rest of the package uses)."""
import ctypes
import os
import threading

import torch

from .. import _lib, flags
from ._base import _p, _stream, _written, _ws, overflow_flag


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
