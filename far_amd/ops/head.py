"""far_amd.ops.head: K2 (bilinear dual-softmax attention) + contraction, K4 solver front end, K11 pose packaging, K15 row-wise dense layers, K12 correlation-volume warp (one family of the torch-tensor front ends for the C ABI in include/far_hip.h; far_amd/ops/__init__.py
re-exports everything under the flat far_amd.ops namespace the rest of the package uses)."""
import ctypes
import os
import threading

import torch

from .. import _lib, flags
from ._base import _p, _stream, _ws, overflow_flag
from .coarse import dual_softmax_stats


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
