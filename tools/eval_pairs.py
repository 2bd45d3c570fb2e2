#!/usr/bin/env python
"""The accuracy clause of north_star, one command away: walk a list of image pairs with ground-truth relative poses through the
product path (far_amd.pipeline.test_step = PL_LoFTR.test_step, lightning_loftr.py:325-343) and print the reference's result table
(test_epoch_end -> aggregate_metrics, lightning_loftr.py:464-547, src/utils/metrics.py:339-377): rotation / translation error means,
medians and percentages, pose AUC@5/10/20, epipolar precision -- for each requested minimal solver side by side (8 = the normalized
8-point north_star names, 5 = Nister's five-point, the solver class the reference executes).

    python tools/eval_pairs.py --ckpt far_8pt.ckpt --pairs mp3d_test_pairs.npz [--minimal 8 5] [--batch 32] [--hyp 2048] [--out table.json]

`--pairs` is an .npz with
    image0, image1   (N, H, W) or (N, 1, H, W), uint8 or float in [0, 1]  -- OR --  paths0, paths1  (N,) image files (read as
                     640 x 480 grayscale the way mp3d_loftr/demo.py does; --root is prepended)
    K0, K1           (N, 3, 3) or (3, 3) intrinsics at the evaluated resolution
    T_0to1           (N, 4, 4) ground-truth relative pose
    identifiers      optional (N,) strings (default pair<i>)
What is needed offline and absent from this container: the checkpoint (`far_8pt.ckpt`, README of the reference) and the Matterport
test split (mp3d_loftr/scripts/eval_matterport.sh:27-37); without --ckpt the seeded synthetic weights run (plumbing only: the table
is then meaningless as accuracy).  tests/test_eval_gpu.py runs this tool on a synthetic pair list and pins its table.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_pairs(path, root=''):
    z = np.load(path, allow_pickle=False)
    n = len(z['T_0to1'])

    def images(key, pkey):
        if key in z:
            a = z[key]
            a = a[:, None] if a.ndim == 3 else a
            return a.astype(np.float32) / 255.0 if a.dtype == np.uint8 else a.astype(np.float32)
        from demo import load_gray
        return np.stack([load_gray(os.path.join(root, str(p)), 640, 480) for p in z[pkey]])[:, None].astype(np.float32)
    K = lambda k: np.broadcast_to(z[k].astype(np.float64), (n, 3, 3)).copy()
    ids = [str(s) for s in z['identifiers']] if 'identifiers' in z else [f'pair{i}' for i in range(n)]
    return {'image0': images('image0', 'paths0'), 'image1': images('image1', 'paths1'), 'K0': K('K0'), 'K1': K('K1'),
            'T_0to1': z['T_0to1'].astype(np.float64), 'identifiers': ids}


def evaluate(model, pairs, minimal=8, batch=32, hyp=2048, seed=0, solver=None, device='cuda'):
    """-> (table of the blended FAR pose, table of the solver's pose alone, per-pair errors).  Tables: aggregate_metrics' dict."""
    from far_amd import metrics as fm
    from far_amd.config import RunCfg
    from far_amd.pipeline import compute_metrics, test_step
    cfg = RunCfg(solver or model.config['solver'], model.config.get('fine_pred_steps', 2), minimal_solver=minimal)
    n = len(pairs['T_0to1'])
    acc = {k: [] for k in ('identifiers', 'epi_errs', 'R_errs', 't_errs', 't_errs_abs', 'successful_fits')}
    acc_s = {k: [] for k in acc}
    for s in range(0, n, batch):
        e = min(n, s + batch)
        t = lambda k, dt: torch.from_numpy(np.ascontiguousarray(pairs[k][s:e])).to(device=device, dtype=dt)
        data = {'image0': t('image0', torch.float32), 'image1': t('image1', torch.float32), 'K0': t('K0', torch.float32),
                'K1': t('K1', torch.float32), 'T_0to1': t('T_0to1', torch.float64), 'dataset_name': ['mp3d'],
                'pair_names': [tuple(pairs['identifiers'][s:e])]}
        test_step(model, data, run_cfg=cfg, H=hyp, seed=seed)
        ret, _ = compute_metrics(data, cfg, H=hyp, seed=seed)                 # the head's blended pose (regressed_rt), as the reference
        for k in acc:
            acc[k] += list(ret['metrics'][k])
        # the solver's own pose of the LAST round (loftr_rt), through the same error function
        rt = data['loftr_rt'].reshape(-1, 3, 4).double()
        te, Re, ta = fm.relative_pose_error_batch(data['T_0to1'], rt[:, :, :3], rt[:, :, 3])
        acc_s['identifiers'] += list(ret['metrics']['identifiers'])
        acc_s['epi_errs'] += list(ret['metrics']['epi_errs'])
        acc_s['R_errs'] += Re.cpu().tolist()
        acc_s['t_errs'] += te.cpu().tolist()
        acc_s['t_errs_abs'] += ta.cpu().tolist()
        acc_s['successful_fits'] += [int(x) for x in data['solver_status'].cpu().tolist()]
    return fm.aggregate_metrics(acc), fm.aggregate_metrics(acc_s), acc


def print_table(title, tab):
    """The lines test_epoch_end prints (lightning_loftr.py:483-492): pose summary first, then the AUCs."""
    print(f'== {title}')
    for k, v in tab.items():
        if 'tr' in k or 'rot' in k or 'pct' in k or 'dset size' in k:
            print(f'{k} {v}')
    print('')
    for k, v in tab.items():
        if 'auc' in k or 'prec' in k:
            print(f'{k} {v}')


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--pairs', required=True)
    ap.add_argument('--ckpt', default=None, help='Lightning checkpoint (keys under matcher.*); default: seeded synthetic weights (plumbing only)')
    ap.add_argument('--root', default='', help='prefix of the image paths in --pairs')
    ap.add_argument('--minimal', type=int, nargs='+', default=[8, 5], choices=[8, 5])
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--hyp', type=int, default=2048)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--precision', default='fp32', help="LoFTR.set_precision mode (default fp32 = the parity configuration)")
    ap.add_argument('--out', default=None, help='write the tables as JSON')
    a = ap.parse_args()
    if not torch.cuda.is_available():
        sys.exit('tools/eval_pairs.py needs a GPU (far_amd has no CPU path)')
    from far_amd import synth
    from far_amd.config import far_eval_config
    from far_amd.loftr import LoFTR
    model = LoFTR(far_eval_config()).eval()
    if a.ckpt:
        sd = torch.load(a.ckpt, map_location='cpu')
        model.load_state_dict(sd.get('state_dict', sd))               # 'matcher.' prefix stripped by LoFTR.load_state_dict
    else:
        print('# no --ckpt: seeded synthetic weights -- the table below checks plumbing, not accuracy', file=sys.stderr)
        synth.load_synthetic(model, seed=0)
    model = model.cuda().set_precision(a.precision)
    pairs = load_pairs(a.pairs, a.root)
    out = {}
    for m in a.minimal:
        far, sol, _ = evaluate(model, pairs, minimal=m, batch=a.batch, hyp=a.hyp, seed=a.seed)
        print_table(f'minimal solver {m}: FAR pose (solver + head, blended)', far)
        print_table(f'minimal solver {m}: solver pose alone (last round)', sol)
        out[f'minimal_{m}'] = {'far': {k: float(v) for k, v in far.items()}, 'solver': {k: float(v) for k, v in sol.items()}}
    if a.out:
        json.dump(out, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
