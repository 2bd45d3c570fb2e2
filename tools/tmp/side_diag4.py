import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from far_amd import synth, ops
import far_amd.ops.head as OH
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
m = LoFTR(far_eval_config()).eval(); synth.load_synthetic(m, seed=0); m = m.cuda()
def batch(n, seed):
    im0, im1 = synth.synth_image_pair(n, seed=seed)
    K = torch.from_numpy(np.stack([synth.MP3D_K] * n)).cuda()
    return {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}
KEEP = []
ws0 = OH._ws
def ws_keep(nb, dev):
    t = ws0(nb, dev)
    if nb > 4_000_000 and nb < 5_000_000:
        KEEP.append(t)
    return t
OH._ws = ws_keep
def run(head):
    m.head_side_stream = head
    del KEEP[:]
    with torch.no_grad():
        d = batch(8, 21)
        d['_far_head_follows'] = True
        m(d); d.pop('_far_head_follows')
        torch.cuda.synchronize()
        f = d[m._HEAD_KEY][1]
        return {'feats': f.feats.clone(), 'enc0': f.enc0.clone()}, [t.clone() for t in KEEP]
a, wa = run(False)
print('workspaces kept', [t.numel() for t in wa])
for i in range(3):
    b, wb = run(True)
    pa, pb = wa[0].view(torch.float32).view(140, 8, 1024), wb[0].view(torch.float32).view(140, 8, 1024)
    df = pa != pb
    print('enc0 equal', torch.equal(a['enc0'], b['enc0']), '| partial diffs', int(df.sum()), 'slices', df.any(2).any(1).nonzero().flatten().tolist()[:40],
          'rows', df.any(2).any(0).nonzero().flatten().tolist(), 'ncols', int(df.any(0).any(0).sum()))
    y = pb.sum(0)
    print('   reduce of the kept (side) partials vs side result:', float((y[:, :512] - b['enc0']).abs().max()), ' vs main result:', float((y[:, :512] - a['enc0']).abs().max()))
head = m.loftr_regress
W = torch.cat([head.encoder[0].weight, head.moe_predictor[0].weight[:, :head.H]], 0).double()
ref = (a['feats'].double() @ W.T)[:, :512]
print('main  vs float64 reference: max err', float((a['enc0'].double() - ref).abs().max()))
print('side  vs float64 reference: max err', float((b['enc0'].double() - ref).abs().max()))
e = (b['enc0'].double() - ref).abs()
print('side err by row', e.max(1).values.tolist())
e = (a['enc0'].double() - ref).abs()
print('main err by row', e.max(1).values.tolist())
