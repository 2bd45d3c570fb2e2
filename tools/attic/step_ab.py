"""Same-box A/B of the full 32-pair step under far_set_tuning knobs: `step_ab.py key=value[,key=value] ...` -- each argument
is one configuration ('base' = no knobs); configurations are timed in interleaved rounds (box-to-box spread is +-2 %, larger
than most single-kernel changes)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far_amd import _lib, synth                     # noqa: E402
from far_amd.config import far_eval_config        # noqa: E402
from far_amd.loftr import LoFTR                    # noqa: E402
from far_amd.pipeline import test_step             # noqa: E402

cfgs = sys.argv[1:] or ['base']
lib = _lib.load()
model = LoFTR(far_eval_config()).eval()
synth.load_synthetic(model, seed=0)
model = model.cuda()
im0, im1 = synth.synth_image_pair(32, seed=1234)
K = torch.from_numpy(np.stack([synth.MP3D_K] * 32)).cuda()
base = {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}


def apply(cfg, on):
    if cfg == 'base':
        return
    for kv in cfg.split(','):
        k, v = kv.split('=')
        lib.far_set_tuning(int(k), int(v) if on else 0)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        test_step(model, dict(base), H=2048, seed=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    run(1)
res = {c: [] for c in cfgs}
for rnd in range(4):
    for c in cfgs:
        apply(c, True)
        run(1)
        res[c].append(run(4))
        apply(c, False)
for c in cfgs:
    print(f'{c:20s} ms/step: ' + ' '.join(f'{x:7.2f}' for x in res[c]) + f'   median {np.median(res[c]):7.2f}')
