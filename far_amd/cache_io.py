"""Cached-prediction on-disk format of the `--from_saved_preds` path (BASELINE configs[3]).

Layout (one torch file per pair id, written by mp3d_loftr/src/lightning/lightning_loftr.py:348-393, read by
src/datasets/interiornet_streetlearn.py:108-118):
    <root>/<split>/loftr_preds/<idx>.pt                  (3, 4) solver pose  [R | t]
    <root>/<split>/loftr_num_correspondences/<idx>.pt    scalar / (1,) inlier count
    <root>/<split>/coarse_features/<idx>.pt              (2, 4800, 256) transformer features of image 0 / 1
    <root>/<split>/loftr_fine_correspondences/<idx>.pt   (M, 4) fine matches [x0, y0, x1, y1] (SAVE_CORR, :381-386)
This module writes that layout from a processed batch and reads it back as ONE packed batch (the reference loads
pair by pair at batch size 1), so the head (+ optionally the GPU solver on cached correspondences) runs at batch 256.
"""
import os

import torch

_DIRS = ('loftr_preds', 'loftr_num_correspondences', 'coarse_features', 'loftr_fine_correspondences')


def save_batch(root, split, pair_ids, data):
    """Write one file set per pair from a data dict produced by far_amd.pipeline.test_step (B pairs)."""
    for d in _DIRS:
        os.makedirs(os.path.join(root, split, d), exist_ok=True)
    rt = data['loftr_rt'].detach().cpu().reshape(-1, 3, 4)
    nc = data['num_correspondences'].detach().cpu().reshape(-1)
    f0, f1 = data['featmap0'].detach().cpu(), data['featmap1'].detach().cpu()
    corr = None
    if 'mkpts0_f' in data and 'm_bids' in data:
        corr = torch.cat([data['mkpts0_f'], data['mkpts1_f']], 1).detach().float().cpu()
        bids = data['m_bids'].detach().cpu()
    for b, idx in enumerate(pair_ids):
        if corr is not None:
            torch.save(corr[bids == b].clone(), os.path.join(root, split, 'loftr_fine_correspondences', f'{int(idx)}.pt'))
        torch.save(rt[b].clone(), os.path.join(root, split, 'loftr_preds', f'{int(idx)}.pt'))
        torch.save(nc[b].clone(), os.path.join(root, split, 'loftr_num_correspondences', f'{int(idx)}.pt'))
        torch.save(torch.stack([f0[b], f1[b]]).clone(), os.path.join(root, split, 'coarse_features', f'{int(idx)}.pt'))


def load_batch(root, split, pair_ids, device='cpu', many_thr_defaults=True, correspondences=False):
    """Packed batch for LoFTR.forward_rt_prediction: featmap0/1 (B, 4800, 256), loftr_rt (B, 3, 4), counts (B,).
    correspondences=True also packs the cached fine matches for the GPU solver: mkpts0_f / mkpts1_f (Mtot, 2) concatenated
    in pair order, m_bids (Mtot,), match_counts (B,) -- what far_amd.supervision.spvs_RT consumes."""
    rts, ncs, f0, f1 = [], [], [], []
    corr, cnt = [], []
    for idx in pair_ids:
        if correspondences:
            c = torch.load(os.path.join(root, split, 'loftr_fine_correspondences', f'{int(idx)}.pt')).float().reshape(-1, 4)
            corr.append(c)
            cnt.append(c.shape[0])
        rts.append(torch.load(os.path.join(root, split, 'loftr_preds', f'{int(idx)}.pt')))
        ncs.append(torch.load(os.path.join(root, split, 'loftr_num_correspondences', f'{int(idx)}.pt')).reshape(()))
        fm = torch.load(os.path.join(root, split, 'coarse_features', f'{int(idx)}.pt'))
        f0.append(fm[0])
        f1.append(fm[1])
    nc = torch.stack(ncs).to(device)
    out = {'loftr_rt': torch.stack(rts).to(device), 'num_correspondences': nc,
           'featmap0': torch.stack(f0).to(device).contiguous(), 'featmap1': torch.stack(f1).to(device).contiguous()}
    if correspondences:
        allc = torch.cat(corr, 0).to(device)
        bids = torch.repeat_interleave(torch.arange(len(pair_ids)), torch.tensor(cnt)).to(device)
        out.update({'mkpts0_f': allc[:, :2].contiguous(), 'mkpts1_f': allc[:, 2:].contiguous(), 'm_bids': bids, 'b_ids': bids,
                    'match_counts': torch.tensor(cnt)})
    if many_thr_defaults:
        # the cached format carries one count only; the three extra counts of `use_many_ransac_thr` default to 0
        z = torch.zeros_like(nc)
        out.update({'num_correspondences_before_ransac': nc.clone(), 'inliers_best_tight': z, 'inliers_best_ultra_tight': z.clone()})
    return out
