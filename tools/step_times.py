"""Per-step wall times of the bench step in a fresh process (is the first process on a box slow throughout, or only at first?).
Usage: python tools/step_times.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from far_amd import synth
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
from far_amd.pipeline import test_step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
model = LoFTR(far_eval_config()).eval()
synth.load_synthetic(model, seed=0)
model = model.to(dev)
im0, im1 = synth.synth_image_pair(32, seed=1234)
K = torch.from_numpy(np.stack([synth.MP3D_K] * 32)).to(dev)
base = {'image0': torch.from_numpy(im0).to(dev), 'image1': torch.from_numpy(im1).to(dev), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
ts, cs = [], []
for i in range(n):
    torch.cuda.synchronize()
    t0, c0 = time.perf_counter(), time.process_time()
    test_step(model, dict(base), H=2048, seed=0)
    c1 = time.process_time()
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
    cs.append(1e3 * (c1 - c0))
print('step ms:', ' '.join(f'{t:.1f}' for t in ts))
# host CPU time of the step's Python (process time, all threads): close to the wall time = the host is the bottleneck on this box
print('host cpu ms:', ' '.join(f'{t:.1f}' for t in cs))
print('cpus:', os.cpu_count(), 'load:', os.getloadavg())
