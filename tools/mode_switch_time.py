"""Step time of the precision modes when the model is SWITCHED between them in one process (what bench.py's other_modes does), with and
without the second stream.  python tools/mode_switch_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from far_amd import synth
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
from far_amd.pipeline import test_step
m = LoFTR(far_eval_config()).eval(); synth.load_synthetic(m, seed=0); m = m.cuda()
im0, im1 = synth.synth_image_pair(32, seed=1234)
K = torch.from_numpy(np.stack([synth.MP3D_K] * 32)).cuda()
base = {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
def step():
    d = dict(base); test_step(m, d, H=2048, seed=0); return d
def timed(n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return 1000 * (time.perf_counter() - t) / n
import gc
if os.environ.get('NOGC'):
    gc.collect(); gc.freeze(); gc.disable()
    print('python gc disabled')
for side in (True, True):
    m.head_side_stream = side
    for rep in range(3):
        for mode in ('fp32', 'fp16', 'mixed16', 'fp16-fine', 'fp32'):
            m.set_precision(mode)
            for _ in range(3): step()
            ts = [timed(3) for _ in range(3)]
            print(f'side stream {side!s:5} {mode:10} ms/step {" ".join(f"{t:7.2f}" for t in ts)}   act_exp {m.act_exp}', flush=True)
