import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from far_amd import synth, ops
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
m = LoFTR(far_eval_config()).eval(); synth.load_synthetic(m, seed=0); m = m.cuda()
def batch(n, seed):
    im0, im1 = synth.synth_image_pair(n, seed=seed)
    K = torch.from_numpy(np.stack([synth.MP3D_K] * n)).cuda()
    return {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
big = torch.zeros(1 << 29, device='cuda')      # 2 GiB
def run(head):
    m.head_side_stream = head
    with torch.no_grad():
        d = batch(8, 21)
        d['_far_head_follows'] = True
        m(d); d.pop('_far_head_follows')
        torch.cuda.synchronize()
        f = d[m._HEAD_KEY][1]
        r1 = f.enc0.clone(); torch.cuda.synchronize()
        h1 = f.enc0._base.cpu()[:, :512].clone()
        big.add_(1.0); torch.cuda.synchronize()
        r2 = f.enc0.clone(); torch.cuda.synchronize()
        h2 = f.enc0._base.cpu()[:, :512].clone()
        return r1, r2, h1, h2, r1.cpu(), d
a = run(False)
print('main-mode: r1==r2', torch.equal(a[0], a[1]), 'h1==h2', torch.equal(a[2], a[3]), 'r1==h1', torch.equal(a[0].cpu(), a[2]))
for i in range(3):
    b = run(True)
    print('side-mode: r1==r2', torch.equal(b[0], b[1]), 'h1==h2', torch.equal(b[2], b[3]), 'r1==h1', torch.equal(b[4], b[2]), 'r2==h2', torch.equal(b[1].cpu(), b[3]),
          '| vs main-mode: r1', torch.equal(a[0], b[0]), 'r2', torch.equal(a[1], b[1]), 'h1', torch.equal(a[2], b[2]), 'h2', torch.equal(a[3], b[3]))
