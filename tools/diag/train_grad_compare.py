"""Diagnostic: the G10 training step on the GPU with the HIP training kernels and with the vendor-op forms, every
gradient of GRAD_KEYS against the golden (reference on CPU) -- separates kernel error from conditioning."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far_amd import synth                                     # noqa: E402
from far_amd.config import far_eval_config                   # noqa: E402
from far_amd.loftr import LoFTR                               # noqa: E402
from far_amd.loftr.transformer import LoFTREncoderLayer, CrossAttention   # noqa: E402
from far_amd.losses import coarse_positive_conf               # noqa: E402
from tests.test_training_cpu import _train_helpers, G         # noqa: E402

h = _train_helpers()
g = np.load(os.path.join(G, 'g10_training.npz'))
base = LoFTR(far_eval_config())
synth.load_synthetic(base, seed=0)
base = base.cuda()
im0, im1, ii, jj, rt = h.train_inputs()
real = torch.randint


def cpu_randint(*a, device=None, **k):
    o = real(*a, **k)
    return o if device is None else o.to(device)


def run(hip, double=False):
    m = copy.deepcopy(base)
    LoFTREncoderLayer.hip_training = hip
    CrossAttention.hip_training = hip
    m.coarse_matching.materialize_conf = not hip
    data = {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(),
            'spv_b_ids': torch.zeros(len(ii), dtype=torch.int64).cuda(), 'spv_i_ids': torch.from_numpy(ii).cuda(),
            'spv_j_ids': torch.from_numpy(jj).cuda()}
    m.train()
    torch.manual_seed(123)
    torch.randint = cpu_randint
    m(data, train=True)
    torch.randint = real
    data.update({'loftr_rt': torch.from_numpy(rt).cuda(), 'num_correspondences': torch.tensor([731]).cuda(),
                 'num_correspondences_before_ransac': torch.tensor([1500]).cuda(),
                 'inliers_best_tight': torch.tensor([410]).cuda(), 'inliers_best_ultra_tight': torch.tensor([57]).cuda()})
    m.forward_rt_prediction(data)
    loss = (-torch.log(coarse_positive_conf(data) + 1e-6).mean() + data['expec_f'].pow(2).mean() + data['regressed_rt'].pow(2).sum())
    m.zero_grad()
    loss.backward()
    return {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}


gh = run(True)
gv = run(False)
for n, k in enumerate(h.GRAD_KEYS):
    a, b = gh[k], gv[k]
    print(f'{k:55s} |g| ref {g["grad_norms"][n]:.4e}  hip {abs(a.norm() - g["grad_norms"][n]) / g["grad_norms"][n]:.2e}  '
          f'vendor-gpu {abs(b.norm() - g["grad_norms"][n]) / g["grad_norms"][n]:.2e}  hip-vs-vendor frob {float((a - b).norm() / b.norm()):.2e}')
worst = sorted(((float((gh[k] - gv[k]).norm() / (gv[k].norm() + 1e-30)), k) for k in gh), reverse=True)[:12]
print('largest hip-vs-vendor relative Frobenius differences:')
for r, k in worst:
    print(f'  {r:.3e}  {k}  |g| {float(gv[k].norm()):.3e}')
