"""Torch / vendor-library compositions of far_amd's fused operators (autograd supplies the backward) -- TEST AND BENCHMARK
INFRASTRUCTURE, not part of the product (moved out of far_amd/ in round 6).  far_amd reaches them only through
far_amd._vendor after `far_amd._vendor.install(tests.vendor_ops)` (tests/conftest.py; bench.py --vendor-train; tools/make_goldens.py).

Who uses them: (1) CPU tensors (the golden G10 / G18 tests, the world-size-2 gloo DDP tests: the package itself raises on CPU
tensors); (2) the explicit comparison legs `hip_training = False` / `materialize_conf = True` (bench.py --workload c3 --vendor-train).
Citations as in the kernels they stand in for.
"""
import torch
import torch.nn.functional as F


def linear_attention(q, k, v, nhead, q_mask=None, kv_mask=None, eps=1e-6):
    """linear_attention.py:31-50.  q (N,L,C), k, v (N,S,C) raw projections -> (N,L,C)."""
    N, L, C = q.shape
    S = k.shape[1]
    D = C // nhead
    Q = F.elu(q.view(N, L, nhead, D)) + 1
    K = F.elu(k.view(N, S, nhead, D)) + 1
    V = v.view(N, S, nhead, D)
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    if kv_mask is not None:
        K = K * kv_mask[:, :, None, None]
        V = V * kv_mask[:, :, None, None]
    V = V / S
    KV = torch.einsum("nshd,nshv->nhdv", K, V)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    out = torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S
    return out.reshape(N, L, C)


def conf_matrix(feat_c0, feat_c1, temperature, mask_c0=None, mask_c1=None):
    """coarse_matching.py:101-118 (dual_softmax)."""
    C = feat_c0.shape[-1]
    f0, f1 = feat_c0 / C ** .5, feat_c1 / C ** .5
    sim = torch.einsum("nlc,nsc->nls", f0, f1) / temperature
    if mask_c0 is not None:
        sim = sim.masked_fill(~(mask_c0[..., None] * mask_c1[:, None]).bool(), -1e9)
    return F.softmax(sim, 1) * F.softmax(sim, 2)


def fine_windows(feat_f, b_ids, cell_ids, W, stride):
    """fine_preprocess.py:40-47: unfold + gather (differentiable w.r.t. feat_f)."""
    N, C = feat_f.shape[:2]
    u = F.unfold(feat_f, kernel_size=(W, W), stride=stride, padding=W // 2)        # (N, C*WW, L)
    u = u.view(N, C, W * W, -1).permute(0, 3, 2, 1)                                 # n l ww c
    return u[b_ids, cell_ids]


def fine_expect(feat_f0, feat_f1):
    """fine_matching.py:43-54 -> coords_normalized (M,2), std (M,)."""
    M, WW, C = feat_f0.shape
    W = int(WW ** .5)
    sim = torch.einsum('mc,mrc->mr', feat_f0[:, WW // 2, :], feat_f1)
    heat = torch.softmax(sim / C ** .5, dim=1)
    lin = torch.linspace(-1, 1, W, device=heat.device, dtype=heat.dtype)
    gy, gx = torch.meshgrid(lin, lin, indexing='ij')
    grid = torch.stack([gx.reshape(-1), gy.reshape(-1)], 1)                          # (WW, 2), x fastest
    coords = heat @ grid
    var = heat @ grid ** 2 - coords ** 2
    std = torch.sum(torch.sqrt(torch.clamp(var, min=1e-10)), -1)
    return coords, std


def bilinear_attention(q, k, v, pos, scale):
    """transformer.py:275-292 for one direction.  q, k, v (Z,N,d), pos (N,6) -> (Z, d+6, d+6)."""
    attn = (q @ k.transpose(-2, -1)) * scale
    A = attn.softmax(dim=-1) * attn.softmax(dim=-2)
    vt = torch.cat([v, pos.unsqueeze(0).expand(v.shape[0], -1, -1)], dim=2)
    return (vt.transpose(-2, -1) @ A) @ vt


def encoder_layer(layer, x, source, x_mask=None, source_mask=None, loftr_preds=None):
    """LoFTREncoderLayer.forward as the reference composes it from its modules (transformer.py:44-67): Linear projections,
    LinearAttention, merge + norm1, MLP on cat[x, message] + norm2, residual."""
    bs = x.size(0)
    q = layer.q_proj(x).view(bs, -1, layer.nhead, layer.dim)
    k = layer.k_proj(source).view(bs, -1, layer.nhead, layer.dim)
    v = layer.v_proj(source).view(bs, -1, layer.nhead, layer.dim)
    msg = layer.attention(q, k, v, q_mask=x_mask, kv_mask=source_mask, loftr_preds=loftr_preds)
    msg = layer.merge(msg.view(bs, -1, layer.nhead * layer.dim))
    msg = layer.norm2(layer.mlp(torch.cat([x, layer.norm1(msg)], dim=2)))                             # :61-66
    return x + msg
