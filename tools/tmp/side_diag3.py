import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from far_amd import synth, ops
from far_amd.config import far_eval_config, RunCfg
from far_amd.loftr import LoFTR
import far_amd.loftr.model as M
m = LoFTR(far_eval_config()).eval(); synth.load_synthetic(m, seed=0); m = m.cuda()
def batch(n, seed):
    im0, im1 = synth.synth_image_pair(n, seed=seed)
    K = torch.from_numpy(np.stack([synth.MP3D_K] * n)).cuda()
    return {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
def run(head):
    m.head_side_stream = head
    with torch.no_grad():
        d = batch(8, 21)
        d['_far_head_follows'] = True
        m(d); d.pop('_far_head_follows')
        torch.cuda.synchronize()
        f = d[m._HEAD_KEY][1]
        return {'feats': f.feats.clone(), 'enc0': f.enc0.clone(), 'mk1': d['mkpts1_f'].clone()}
a = run(False)
def cmp(tag):
    for i in range(2):
        b = run(True)
        print(tag, {k: bool(torch.equal(a[k], b[k])) for k in a})
cmp('plain side stream       ')
orig = LoFTR._head_features
def synced(self, *ar, **kw):
    if kw.get('join') is False:
        torch.cuda.synchronize()
    r = orig(self, *ar, **kw)
    if kw.get('join') is False:
        torch.cuda.synchronize()
    return r
LoFTR._head_features = synced
cmp('side stream, serialised ')
def synced_after(self, *ar, **kw):
    r = orig(self, *ar, **kw)
    if kw.get('join') is False:
        torch.cuda.synchronize()
    return r
LoFTR._head_features = synced_after
cmp('side, sync after only   ')
def norec(self, *ar, **kw):
    kw2 = dict(kw)
    if kw.get('join') is False:
        rs = torch.Tensor.record_stream
        torch.Tensor.record_stream = lambda *a_, **k_: None
        try:
            return orig(self, *ar, **kw2)
        finally:
            torch.Tensor.record_stream = rs
    return orig(self, *ar, **kw2)
LoFTR._head_features = norec
cmp('side, no record_stream  ')
LoFTR._head_features = orig
# the stage on the side stream but rows_linear patched to run on the main stream afterwards? -> instead: a sync in front of rows_linear
import far_amd.loftr.transformer as T
rl = ops.rows_linear
def rl_sync(x, pr, *ar, **kw):
    if x.shape[1] > 2048:
        torch.cuda.synchronize()
    return rl(x, pr, *ar, **kw)
ops.rows_linear = rl_sync
cmp('side, sync before K15   ')
ops.rows_linear = rl
