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
if len(sys.argv) > 1 and sys.argv[1] == 'train':
    # the training step (BASELINE configs[2] shape: one pair): forward + backward + AdamW with the collector disabled
    import copy
    from far_amd.config import far_train_config, RunCfg
    from far_amd.losses import LoFTRLoss
    from far_amd.pipeline import train_step
    mt = copy.deepcopy(m).train()
    loss_fn = LoFTRLoss(far_train_config()).train()
    opt = torch.optim.AdamW(mt.parameters(), lr=1e-5)
    base = synth.synth_training_batch(1, seed=78, device='cuda')

    def tstep():
        batch = dict(base)
        train_step(mt, batch, loss_fn, RunCfg('prior_ransac', 2), H=256, seed=0)
        batch['loss'].backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
    for _ in range(3):
        tstep()
    gc.collect(); gc.disable()
    mem = []
    for i in range(10):
        tstep()
        mem.append(torch.cuda.memory_allocated() / 2**30)
    print('training step: allocated after each step (GiB), collector disabled:', ' '.join(f'{x:.3f}' for x in mem))
    print('unreachable objects found by a collection now:', gc.collect())
    sys.exit(0)
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
