#!/usr/bin/env python
"""demo.py -- counterpart of mp3d_loftr/demo.py (BASELINE configs[0]): relative pose of ONE image pair through
matcher -> solver -> head -> solver(prior) -> head, printing the solver pose of the last round (demo.py:145-151 prints
batch['loftr_rt']).

    python demo.py --img_path0 a.png --img_path1 b.png [--ckpt_path far_8pt.ckpt] [--fx 517.97 --fy 517.97 --cx 320 --cy 240]
    python demo.py --synthetic                         # the seeded synthetic pair + synthetic checkpoint (no files needed)

The product path needs the MI355X: there is no CPU fallback (far_amd/_lib.py, far_amd/ops.py raise), so on a machine
without a GPU this script says so and exits with status 2 instead of computing something else.  `--check` additionally
runs the CPU oracle (oracle/model.py, test infrastructure) on the same pair AS A CHECKER and reports the deviation of the
matcher stage; it never feeds the product path.  Images: 8-bit grayscale through Pillow (the reference uses cv2.imread +
cv2.resize; bilinear resize here, so pixel values -- hence poses -- of RESIZED inputs may differ in the last bits; images
already at --w x --h are read identically).
"""
import argparse
import sys

import numpy as np
import torch


def load_gray(path, w, h):
    from PIL import Image
    im = Image.open(path).convert('L')
    if im.size != (w, h):
        im = im.resize((w, h), Image.BILINEAR)
    return torch.from_numpy(np.asarray(im, dtype=np.float32))[None, None] / 255


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--img_path0')
    ap.add_argument('--img_path1')
    ap.add_argument('--ckpt_path', default=None, help='Lightning checkpoint (keys under matcher.*); default: seeded synthetic weights')
    ap.add_argument('--w', type=int, default=640)
    ap.add_argument('--h', type=int, default=480)
    ap.add_argument('--fx', type=float, default=517.97)
    ap.add_argument('--fy', type=float, default=517.97)
    ap.add_argument('--cx', type=float, default=320.0)
    ap.add_argument('--cy', type=float, default=240.0)
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--check', action='store_true', help='also run the CPU oracle on the pair and report the deviation')
    a = ap.parse_args()
    if not torch.cuda.is_available():
        print('demo.py: no GPU visible -- far_amd has no CPU path (the HIP kernels ARE the product); run it on the MI355X box',
              file=sys.stderr)
        return 2
    from far_amd import synth
    from far_amd.config import far_eval_config
    from far_amd.loftr import LoFTR
    from far_amd.pipeline import test_step
    cfg = far_eval_config()
    model = LoFTR(cfg).eval()
    if a.ckpt_path:
        sd = torch.load(a.ckpt_path, map_location='cpu')
        model.load_state_dict(sd.get('state_dict', sd))                 # 'matcher.' prefix stripped by LoFTR.load_state_dict
    else:
        synth.load_synthetic(model, seed=0)
    model = model.cuda()
    if a.synthetic or not a.img_path0:
        im0, im1 = synth.synth_image_pair(1, seed=7, hw=(a.h, a.w))
        im0, im1 = torch.from_numpy(im0), torch.from_numpy(im1)
    else:
        im0, im1 = load_gray(a.img_path0, a.w, a.h), load_gray(a.img_path1, a.w, a.h)
    K = torch.tensor([[[a.fx, 0, a.cx], [0, a.fy, a.cy], [0, 0, 1]]], dtype=torch.float64).cuda()
    batch = {'image0': im0.cuda(), 'image1': im1.cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d'], 'pair_id': 0,
             'pair_names': (a.img_path0, a.img_path1)}
    test_step(model, batch)
    print('matches:', int(batch['b_ids'].numel()), ' inliers:', int(batch['num_correspondences'][0]))
    print('predicted pose is:\n', np.round(batch['loftr_rt'].cpu().numpy(), 4))
    reg = batch['regressed_rt'][0].float().cpu().numpy()
    print('regressed (normalised 6D) pose:', np.round(reg, 4))
    if a.check:
        import json
        import os
        from oracle import model as om
        root = os.path.dirname(os.path.abspath(__file__))
        if a.ckpt_path:
            w = om.Weights({k[len('matcher.'):] if k.startswith('matcher.') else k: v for k, v in sd.get('state_dict', sd).items()})
        else:
            man = json.load(open(os.path.join(root, 'tests', 'golden', 'g8_state_dict_manifest.json')))
            w = om.Weights(synth.synthetic_state_dict({k: tuple(v) for k, v in man.items()}))
        od = om.matcher_forward(w, cfg, im0.numpy(), im1.numpy())
        got = set(zip(batch['i_ids'].tolist(), batch['j_ids'].tolist()))
        ref = set(zip(od['i_ids'].tolist(), od['j_ids'].tolist()))
        print(f'oracle check: {len(got & ref)} of {len(ref)} oracle matches reproduced ({len(got)} found); '
              f'max |featmap0 - oracle| = {np.abs(batch["featmap0"].cpu().numpy() - od["featmap0"]).max():.2e}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
