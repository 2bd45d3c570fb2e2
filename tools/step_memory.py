"""Does a step leave device memory behind that only Python's cyclic collector frees?  Ten steps with the collector DISABLED:
torch.cuda.memory_allocated() must stay flat from step to step (round 6: two self-referential closures kept every step's data dict --
gigabytes of device tensors -- alive until a generation-2 collection; the pool grew to 270 GB without one).
python tools/step_memory.py"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from far_amd import synth
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
from far_amd.pipeline import test_step
m = LoFTR(far_eval_config()).eval(); synth.load_synthetic(m, seed=0); m = m.cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
im0, im1 = synth.synth_image_pair(n, seed=1234)
K = torch.from_numpy(np.stack([synth.MP3D_K] * n)).cuda()
base = {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
for _ in range(3):
    test_step(m, dict(base), H=2048, seed=0)
torch.cuda.synchronize(); gc.collect(); gc.disable()
mem = []
for i in range(10):
    test_step(m, dict(base), H=2048, seed=0)
    torch.cuda.synchronize()
    mem.append(torch.cuda.memory_allocated() / 2**30)
print('allocated after each step (GiB), collector disabled:', ' '.join(f'{x:.2f}' for x in mem))
print('reserved (GiB):', round(torch.cuda.memory_reserved() / 2**30, 2), ' unreachable objects found by a collection now:', gc.collect())
assert mem[-1] - mem[0] < 0.05, 'device memory grows from step to step without the cyclic collector'
print('flat')
