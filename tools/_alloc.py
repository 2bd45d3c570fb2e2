import sys, time; sys.path.insert(0,'.')
import torch, numpy as np
from far_amd import synth
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
from far_amd.pipeline import test_step
m = LoFTR(far_eval_config()).eval(); synth.load_synthetic(m, seed=0); m = m.cuda()
im0, im1 = synth.synth_image_pair(32, seed=1234)
K = torch.from_numpy(np.stack([synth.MP3D_K] * 32)).cuda()
base = {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
for i in range(4):
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = dict(base); test_step(m, b, H=2048, seed=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1 = torch.cuda.memory_stats()
    print(i, f'{dt*1000:.1f} ms', 'device_alloc', s1['num_device_alloc'] - s0['num_device_alloc'], 'device_free', s1['num_device_free'] - s0['num_device_free'],
          'retries', s1['num_alloc_retries'] - s0['num_alloc_retries'], 'reserved GB', s1['reserved_bytes.all.current'] / 2**30, 'peak alloc GB', s1['allocated_bytes.all.peak'] / 2**30, flush=True)
