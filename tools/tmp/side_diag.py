import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from far_amd import synth
from far_amd.config import far_eval_config, RunCfg
from far_amd.loftr import LoFTR
from far_amd.pipeline import test_step
from far_amd.supervision import compute_supervision_RT
m = LoFTR(far_eval_config()).eval(); synth.load_synthetic(m, seed=0); m = m.cuda()
def batch(n, seed):
    im0, im1 = synth.synth_image_pair(n, seed=seed)
    K = torch.from_numpy(np.stack([synth.MP3D_K] * n)).cuda()
    return {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
cfg = RunCfg(m.config['solver'], 2)
def run(head):
    m.head_side_stream = head
    out = {}
    with torch.no_grad():
        d = batch(8, 21)
        d['_far_head_follows'] = True
        m(d); d.pop('_far_head_follows')
        torch.cuda.synchronize()
        f = d[m._HEAD_KEY][1]
        out['feats'], out['enc0'], out['moe0'] = f.feats.clone(), f.enc0.clone(), f.moe0.clone()
        out['mk1'] = d['mkpts1_f'].clone()
        d['translation_scale'] = None
        compute_supervision_RT(d, cfg, H=256, seed=0)
        torch.cuda.synchronize()
        out['rt1'] = d['loftr_rt'].clone()
        m.forward_rt_prediction(d)
        torch.cuda.synchronize()
        out['reg1'] = d['regressed_rt'].clone()
        compute_supervision_RT(d, cfg, H=256, seed=0)
        out['rt2'] = d['loftr_rt'].clone()
        m.forward_rt_prediction(d)
        out['reg2'] = d['regressed_rt'].clone()
        torch.cuda.synchronize()
    return out
a = run(False); a2 = run(False)
for k in a: print('off vs off', k, torch.equal(a[k], a2[k]))
for i in range(3):
    b = run(True)
    for k in a: print('off vs on ', k, torch.equal(a[k], b[k]), float((a[k].double() - b[k].double()).abs().max()))
d1 = batch(8, 21); d2 = batch(8, 21)
m.head_side_stream = False; test_step(m, d1, H=256); m.head_side_stream = True; test_step(m, d2, H=256); torch.cuda.synchronize()
for k in ('mkpts1_f', 'loftr_rt', 'regressed_rt'): print('test_step', k, torch.equal(d1[k], d2[k]))
print('--- pattern of differences in enc0|moe0')
for i in range(3):
    b = run(True)
    for k in ('enc0', 'moe0'):
        df = (a[k] != b[k])
        rows = df.any(1).nonzero().flatten().tolist()
        cols = df.any(0).nonzero().flatten()
        print(k, 'shape', tuple(a[k].shape), 'n diff', int(df.sum()), 'rows', rows, 'cols min/max/n', int(cols.min()) if len(cols) else None, int(cols.max()) if len(cols) else None, len(cols),
              'typical |a|', float(a[k].abs().mean()))
df = (a['enc0'] != b['enc0'])
print('cols', df.any(0).nonzero().flatten().tolist())
print('per row counts', df.sum(1).tolist())
print('diff sample', (a['enc0'] - b['enc0'])[0][df[0]][:16].tolist())
