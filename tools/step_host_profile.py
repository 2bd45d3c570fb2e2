"""Host-side view of the headline step (cProfile over 10 steps): which Python functions the 25-30 ms of host time per step go to.
The GPU has ~87 ms of work per step, most of it queued ahead of the one host read (the match count); what the host does AFTER that
read is what can starve the GPU on a slow box.  Usage: python tools/step_host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from far_amd import synth
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
from far_amd.pipeline import test_step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda', 0)
model = LoFTR(far_eval_config()).eval()
synth.load_synthetic(model, seed=0)
model = model.to(dev)
im0, im1 = synth.synth_image_pair(32, seed=1234)
K = torch.from_numpy(np.stack([synth.MP3D_K] * 32)).to(dev)
base = {'image0': torch.from_numpy(im0).to(dev), 'image1': torch.from_numpy(im1).to(dev), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
for _ in range(4):
    test_step(model, dict(base), H=2048, seed=0)
torch.cuda.synchronize()
# host time of a step when the GPU is NOT waited for: enqueue-only time up to the first blocking read, then the rest
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    test_step(model, dict(base), H=2048, seed=0)
pr.disable()
torch.cuda.synchronize()
print(f'wall per step {1e3 * (time.perf_counter() - t0) / n:.1f} ms (profiled: slower than plain)')
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(30)
