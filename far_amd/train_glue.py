"""GPU-training autograd compositions of the two operators that have no backward kernel (accepted, listed in DESIGN.md section 10):
the fine-window expectation (fine_matching.py:43-54; 25 x 25 values per match) and the dense confidence matrix of PADDED-MASK batches
(coarse_matching.py:101-118 with masks; the sparse-position kernels of K1 carry no masks).  Small elementwise / reduction tensor
methods on GPU tensors; everything else of the training step runs HIP forward + backward kernels."""
import torch


def masked_conf_matrix(feat_c0, feat_c1, temperature, mask_c0, mask_c1):
    """coarse_matching.py:101-118 (dual_softmax) with the padded-mask fill (:110-117)."""
    C = feat_c0.shape[-1]
    sim = torch.einsum("nlc,nsc->nls", feat_c0 / C ** .5, feat_c1 / C ** .5) / temperature
    if mask_c0 is not None:
        sim = sim.masked_fill(~(mask_c0[..., None] * mask_c1[:, None]).bool(), -1e9)
    return sim.softmax(1) * sim.softmax(2)


def fine_expect(feat_f0, feat_f1):
    """fine_matching.py:43-54 -> coords_normalized (M,2), std (M,)."""
    M, WW, C = feat_f0.shape
    W = int(WW ** .5)
    sim = torch.einsum('mc,mrc->mr', feat_f0[:, WW // 2, :], feat_f1)
    heat = (sim / C ** .5).softmax(1)
    lin = torch.linspace(-1, 1, W, device=heat.device, dtype=heat.dtype)
    gy, gx = torch.meshgrid(lin, lin, indexing='ij')
    grid = torch.stack([gx.reshape(-1), gy.reshape(-1)], 1)                          # (WW, 2), x fastest
    coords = heat @ grid
    var = heat @ grid ** 2 - coords ** 2
    std = torch.sum(torch.sqrt(torch.clamp(var, min=1e-10)), -1)
    return coords, std
