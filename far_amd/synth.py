"""Seeded synthetic checkpoint and synthetic image pairs (no network, no datasets, no pretrained weights).

Used by bench.py, __graft_entry__.smoke(), the tests and tools/make_goldens.py so that all of them build the
identical 51.09 M parameters from a seed instead of shipping them.  Plain numpy; nothing here is a kernel.

A purely random LoFTR produces a degenerate dual-softmax (ReLU features share a large common component, one
column wins every row, M = 0..1).  Two adjustments make the synthetic model behave like a trained one on
synthetic pairs (M ~ 1.9 k confident, geometrically consistent matches per 640x480 pair):
  * encoder-layer norm2 gains are scaled by 0.2 (messages stay small next to the residual stream);
  * `backbone.layer3_outconv.weight` is replaced by a calibrated tensor whose rows are orthogonal to the
    mean activation of its input and whose output has unit spatial std
    (far_amd/assets/synth_layer3_outconv_s0.npy, made by `python -m far_amd.synth --calibrate`).
"""
import os
import zlib

import numpy as np

_ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', 'synth_layer3_outconv_s{seed}.npy')


def _raw_state_dict(shapes, seed=0):
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            a = np.zeros(shp, np.int64)
        elif leaf == 'running_mean':
            a = (0.05 * rng.standard_normal(shp)).astype(np.float32)
        elif leaf == 'running_var':
            a = rng.uniform(0.8, 1.2, shp).astype(np.float32)
        elif len(shp) <= 1:
            is_norm = ('norm' in name or 'bn1' in name or 'bn2' in name or 'downsample.1' in name
                       or ('outconv2.1.' in name))
            if leaf == 'weight' and is_norm:
                a = rng.uniform(0.9, 1.1, shp).astype(np.float32)
            else:
                a = (0.02 * rng.standard_normal(shp)).astype(np.float32)
        elif name.endswith('pos_embed'):
            a = (0.02 * rng.standard_normal(shp)).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = (rng.standard_normal(shp) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        if leaf == 'weight' and 'norm2' in name and '.layers.' in name:
            a = (a * np.float32(0.2)).astype(np.float32)
        out[name] = a
    return out


def synthetic_state_dict(shapes, seed=0, calibrated=True):
    """{name: np.ndarray} for a {name: shape} manifest (e.g. {k: v.shape for k, v in model.state_dict().items()})."""
    sd = _raw_state_dict(shapes, seed)
    key = 'backbone.layer3_outconv.weight'
    if calibrated and key in sd:
        path = _ASSET.format(seed=seed)
        if not os.path.exists(path):
            raise FileNotFoundError(f'{path} missing: run `python -m far_amd.synth --calibrate --seed {seed}`')
        sd[key] = np.load(path).astype(np.float32).reshape(sd[key].shape)
    return sd


def load_synthetic(model, seed=0):
    """Load the synthetic checkpoint into a LoFTR-shaped torch module (reference's or far_amd's)."""
    import torch
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synthetic_state_dict(shapes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model


def synth_image_pair(N, seed=0, hw=(480, 640), disparities=(8, 40, 72), noise=0.01):
    """N grayscale pairs in [0,1], float32 (N,1,H,W).  image1 is image0 displaced in x by a different
    disparity in each horizontal band (a lateral camera translation in front of a few depth planes), so the
    two-view geometry is well posed; disparities are multiples of the coarse cell (8 px)."""
    rng = np.random.default_rng(seed)
    H, W = hw
    dmax = max(disparities)
    base = rng.random((N, 1, H // 8 + 2, (W + dmax) // 8 + 2)).astype(np.float32)
    big = np.kron(base, np.ones((1, 1, 8, 8), np.float32))
    fine = rng.random((N, 1, big.shape[2], big.shape[3])).astype(np.float32)
    tex = (0.7 * big + 0.3 * fine)[:, :, :H, :W + dmax]
    im0 = tex[:, :, :, :W].copy()
    im1 = np.empty_like(im0)
    nb = len(disparities)
    for k, d in enumerate(disparities):
        r0, r1 = (H * k) // nb // 8 * 8, (H * (k + 1)) // nb // 8 * 8 if k < nb - 1 else H
        im1[:, :, r0:r1, :] = tex[:, :, r0:r1, d:d + W]
    im1 = np.clip(im1 + noise * rng.standard_normal(im1.shape), 0, 1).astype(np.float32)
    return im0, im1


MP3D_K = np.array([[517.97, 0, 320.], [0, 517.97, 240.], [0, 0, 1.]])   # src/utils/dataset.py:201-211


def synth_training_batch(B, seed=0, device='cpu', disparities=(8, 40, 72)):
    """A training batch for the 640x480 banded pairs of synth_image_pair WITH its supervision, in the form the
    reference's spvs_coarse leaves in the data dict (supervision.py:122-137): ground-truth coarse matches (spv_*_ids),
    the warped coarse grid of image 0 and the grid of image 1 (for spvs_fine), the relative pose (lateral translation)."""
    import torch
    im0, im1 = synth_image_pair(B, seed=seed, disparities=disparities)
    ys, xs = np.meshgrid(np.arange(60), np.arange(80), indexing='ij')
    nb = len(disparities)
    band = np.array([min(nb - 1, next(k for k in range(nb) if y * 8 < ((480 * (k + 1)) // nb // 8 * 8 if k < nb - 1 else 480)))
                     for y in range(60)])
    d_c = (np.array(disparities)[band] // 8)[:, None] + 0 * xs
    ok = xs - d_c >= 0
    ii = (ys * 80 + xs)[ok].astype(np.int64)
    jj = (ys * 80 + xs - d_c)[ok].astype(np.int64)
    grid = (np.stack([xs, ys], -1).reshape(1, 4800, 2) * 8).astype(np.float32)
    w_pt0 = grid - np.stack([8 * d_c, 0 * d_c], -1).reshape(1, 4800, 2).astype(np.float32)
    T = np.eye(4, dtype=np.float32)
    T[0, 3] = -1.0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    K = t(np.stack([MP3D_K] * B))
    return {'image0': t(im0), 'image1': t(im1), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d'],
            'T_0to1': t(T)[None].repeat(B, 1, 1),
            'spv_b_ids': torch.arange(B, device=device).repeat_interleave(len(ii)), 'spv_i_ids': t(ii).repeat(B),
            'spv_j_ids': t(jj).repeat(B), 'spv_w_pt0_i': t(w_pt0).repeat(B, 1, 1), 'spv_pt1_i': t(grid).repeat(B, 1, 1)}


def calibrate(seed=0):
    """Compute the calibrated layer3_outconv weight with far_amd's own backbone on CPU and store the asset."""
    import torch
    from .config import far_eval_config
    from .loftr.backbone import build_backbone
    bb = build_backbone(far_eval_config()).eval()
    shapes = {'backbone.' + k: tuple(v.shape) for k, v in bb.state_dict().items()}
    sd = _raw_state_dict(shapes, seed)
    bb.load_state_dict({k[len('backbone.'):]: torch.from_numpy(v) for k, v in sd.items()})
    c0, c1 = synth_image_pair(2, seed=999, disparities=(40,))
    with torch.no_grad():
        x = torch.from_numpy(np.concatenate([c0, c1]))
        x3 = bb.layer3(bb.layer2(bb.layer1(bb.relu(bb.bn1(bb.conv1(x))))))
        mu = x3.double().mean(dim=(0, 2, 3))
        Wm = bb.layer3_outconv.weight[:, :, 0, 0].double()
        Wc = Wm - (Wm @ mu)[:, None] * mu[None, :] / (mu @ mu)
        out = torch.einsum('ok,nkhw->nohw', Wc, x3.double())
        Wc = (Wc / out.std()).float().numpy()
    os.makedirs(os.path.dirname(_ASSET), exist_ok=True)
    np.save(_ASSET.format(seed=seed), Wc.astype(np.float32))
    return Wc


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--calibrate', action='store_true')
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    if a.calibrate:
        w = calibrate(a.seed)
        print('saved', _ASSET.format(seed=a.seed), w.shape, float(np.abs(w).mean()))
