"""Python side of the parked K21 experiment (tools/experiments/layer128_fused_f16s.hip): the weight image and the op, as they were
wired into far_amd/ops.py while the kernel was built into the library (needs the entry points in include/far_hip.h / _lib.py again)."""
import torch
from far_amd import _lib
from far_amd.ops import PackedAttn, PackedMlp, _p, _stream, _written, overflow_flag


class PackedLayer128:
    """Weight image of far_layer128_f16s (K21, the whole d_model-128 encoder layer): PackedAttn's 16 slabs, then the MLP's 24 in
    the fused kernel's order -- per half hh of the hidden dimension (hidden tiles 4 hh .. 4 hh + 3): four message slabs of GEMM 1
    (message tile ct: [k-step u][hidden tile][plane][lane][8], element e = input channel 128 + 32 ct + 16 u + 4 h + (e & 3) +
    8 (e >> 2) -- the order in which the transposed merge leaves a token's message in the accumulator registers), four x slabs
    (chunk c: element e = input channel 32 c + 16 h + 8 ks + e, as PackedAttn's projections), four GEMM-2 slabs (PackedMlp's)."""

    def __init__(self, wq, wk, wv, wm, w0, w2):
        lib = _lib.load()
        attn, mlp = PackedAttn(wq, wk, wv, wm), PackedMlp(w0, w2)
        d = attn.d
        w0 = w0.detach().float() * 2.0 ** mlp.e0
        dev = w0.device
        ar = lambda n: torch.arange(n, device=dev)
        hh_, c_, k_, t_, l_, e_ = (ar(2).view(2, 1, 1, 1, 1, 1), ar(4).view(1, 4, 1, 1, 1, 1), ar(2).view(1, 1, 2, 1, 1, 1),
                                   ar(4).view(1, 1, 1, 4, 1, 1), ar(64).view(1, 1, 1, 1, 64, 1), ar(8).view(1, 1, 1, 1, 1, 8))
        shp = (2, 4, 2, 4, 64, 8)
        hid = (32 * (4 * hh_ + t_) + (l_ & 31)).expand(shp)
        km = (d + 32 * c_ + 16 * k_ + 4 * (l_ >> 5) + (e_ & 3) + 8 * (e_ >> 2)).expand(shp)      # c_ = message tile ct, k_ = u
        kx = (32 * c_ + 16 * (l_ >> 5) + 8 * k_ + e_).expand(shp)                                # c_ = x chunk, k_ = k-step

        def planes(v):                                     # (hh, c, ks, t, l, e) -> (hh, c, ks, t, plane, l, e)
            hi = v.half()
            return torch.stack([hi, (v - hi.float()).half()], 4)
        gm, gx = planes(w0[hid, km]), planes(w0[hid, kx])
        slab = 16384 // 2
        g2 = mlp.packed.view(torch.float16)[16 * slab:].view(2, 4 * slab)                       # GEMM-2 slabs of hidden tiles 0-3 | 4-7
        parts = [attn.packed.view(torch.float16)]
        for hh in range(2):
            parts += [gm[hh].reshape(-1), gx[hh].reshape(-1), g2[hh]]
        self.packed = torch.cat(parts).view(torch.uint8)
        assert self.packed.numel() == lib.far_layer128_packed_bytes(d)
        self.d, self.scales, self.hscale, self.oscale = d, attn.scales, mlp.hscale, mlp.oscale


def layer128_fused(x, source, pack, nhead, norm1, norm2, attn_eps=1e-6, out=None):
    """K21: LoFTREncoderLayer.forward (transformer.py:44-67) at d_model = 128 on (N, L <= 32, 128) windows in one launch;
    norm1 / norm2 = (gamma, beta, eps)."""
    lib = _lib.load()
    N, L, d = x.shape
    S = source.shape[1]
    if d != pack.d or source.shape[0] != N or source.shape[2] != d:
        raise _lib.FarHipError('layer128_fused: x (N, L, 128) and source (N, S, 128) expected')
    y = torch.empty_like(x) if out is None else out
    sk, sv, sq, sm = pack.scales
    f = lambda t: _p(t, torch.float32)
    rc = lib.far_layer128_f16s(f(x), f(source), _p(pack.packed), N, L, S, d, int(nhead), sk, sv, sq, sm, float(attn_eps),
                               f(norm1[0]), f(norm1[1]), float(norm1[2]), pack.hscale, pack.oscale, f(norm2[0]), f(norm2[1]),
                               float(norm2[2]), f(y), _p(overflow_flag(x.device)), _stream())
    _lib.check(rc, 'far_layer128_f16s')
    return y if out is None else _written(y)


