"""VERDICT r3 item 5 (hi.hi prescreen of K1 / K2 score tiles): how many 64-column score tiles of the bench workload could be skipped at
all?  One bench step (4 pairs), then for the coarse matcher's scores (log2 domain, as K1 computes them) and the head's bilinear
attention scores (K2): the share of (row, 64-column tile) pairs whose largest entry lies more than 36 below the row maximum (what
the statistics passes could skip while keeping 4800 dropped terms below 2^-24 of the sum), and -- match pass -- the share of tiles
holding no entry that can reach the confidence threshold.  Usage: python tools/prescreen_stats.py"""
import math
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from far_amd import synth
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
from far_amd.pipeline import test_step

dev = 'cuda'
cfg = far_eval_config()
model = LoFTR(cfg).eval()
synth.load_synthetic(model, seed=0)
model = model.to(dev)
B = 4
im0, im1 = synth.synth_image_pair(B, seed=1234)
K = torch.from_numpy(np.stack([synth.MP3D_K] * B)).to(dev)
batch = {'image0': torch.from_numpy(im0).to(dev), 'image1': torch.from_numpy(im1).to(dev), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
grabbed = {}
cm = model.coarse_matching
orig = cm.forward


def hook(feat_c0, feat_c1, data, *a, **k):
    grabbed['f0'], grabbed['f1'] = feat_c0.detach().clone(), feat_c1.detach().clone()
    return orig(feat_c0, feat_c1, data, *a, **k)


cm.forward = hook
with torch.no_grad():
    test_step(model, batch, H=2048, seed=0)


def tile_report(name, x, thr_log2=None):
    """x: (Z, L, S) scores in the log2 domain."""
    Z, L, S = x.shape
    nt = S // 64
    t = x[:, :, :nt * 64].reshape(Z, L, nt, 64).amax(-1)                       # tile maxima per row
    rmax = x.amax(-1, keepdim=True)
    skip = (t < rmax - 36.0).float().mean().item()
    gap = (rmax - t).flatten()
    q = torch.quantile(gap[torch.randperm(gap.numel(), device=gap.device)[:2_000_000]], torch.tensor([0.5, 0.9, 0.99], device=gap.device)).tolist()
    line = f'{name}: rows x tiles with tile max < row max - 36: {100 * skip:.2f} %; row max - tile max: median {q[0]:.1f}, 90 % {q[1]:.1f}, 99 % {q[2]:.1f} (log2 units)'
    if thr_log2 is not None:
        cmax = x.amax(-2, keepdim=True)
        can = (2 * x - rmax - cmax >= thr_log2)                                   # conf >= thr needs at least this (row / column sums >= 1)
        tiles = can[:, :, :nt * 64].reshape(Z, L, nt, 64).any(-1)
        w = tiles.reshape(Z, L // 32, 32, nt).any(2)                              # a wave's 32 rows decide together
        line += f'; match pass: 32-row x 64-column tiles with a possible match {100 * w.float().mean().item():.2f} %'
    print(line)


f0, f1 = grabbed['f0'], grabbed['f1']
C = f0.shape[-1]
x = torch.einsum('nlc,nsc->nls', f0, f1) * (math.log2(math.e) / (C * cm.temperature))
tile_report(f'K1 coarse scores ({B} pairs, {x.shape[1]} x {x.shape[2]})', x, thr_log2=math.log2(cm.thr))
# K2: the head's CrossAttention scores -- 4 heads of 64 channels, scale 1 / 8
from far_amd import ops
orig_emm = ops.emm_bilinear_planes


def emm_hook(qkv, pos, scale, Bp):
    if 'q' not in grabbed:                      # qkv (3 h, 2 B, N, 64): direction 0 pairs the queries of image 1 with the keys of image 0
        h = qkv.shape[0] // 3
        grabbed['q'] = qkv[0:h, Bp:2 * Bp].permute(1, 0, 2, 3).reshape(-1, qkv.shape[2], qkv.shape[3]).detach().clone()
        grabbed['k'] = qkv[h:2 * h, 0:Bp].permute(1, 0, 2, 3).reshape(-1, qkv.shape[2], qkv.shape[3]).detach().clone()
    return orig_emm(qkv, pos, scale, Bp)


ops.emm_bilinear_planes = emm_hook
batch2 = {kk: v for kk, v in batch.items() if kk in ('image0', 'image1', 'K0', 'K1', 'dataset_name')}
with torch.no_grad():
    test_step(model, batch2, H=2048, seed=0)
if 'q' in grabbed:
    q, k = grabbed['q'], grabbed['k']
    print('K2 q / k shapes', tuple(q.shape), tuple(k.shape))
    qq = q.reshape(-1, q.shape[-2], q.shape[-1])[:8].float()
    kk = k.reshape(-1, k.shape[-2], k.shape[-1])[:8].float()
    xs = torch.einsum('zld,zsd->zls', qq, kk) * (0.125 * math.log2(math.e))
    tile_report(f'K2 head attention scores (8 of {q.numel() // (q.shape[-2] * q.shape[-1])} problems)', xs)
else:
    print('K2 hook not reached')
