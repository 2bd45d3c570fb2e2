"""GPU probe: speed and deviation of bf16 vendor-path modes vs the fp32 path (not a test; prints numbers)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from far_amd import synth
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
from far_amd.pipeline import test_step

m = LoFTR(far_eval_config()).eval(); synth.load_synthetic(m); m = m.cuda()
N = 32
im0, im1 = synth.synth_image_pair(N, seed=77)
K = torch.from_numpy(np.stack([synth.MP3D_K] * N)).cuda()
base = {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda(), 'K0': K, 'K1': K.clone(), 'dataset_name': ['mp3d']}

def run(dt):
    m.backbone_dtype = dt
    d = dict(base)
    with torch.no_grad():
        for _ in range(2): m.forward_feature_extraction(dict(base))
        torch.cuda.synchronize(); t = time.time()
        m.forward_feature_extraction(d)
        torch.cuda.synchronize(); tb = time.time() - t
        m.forward_correspondence_prediction(d)
    return d, tb

d32, t32 = run(torch.float32)
for fd in [torch.float16, torch.bfloat16]:
    m.backbone.fine_branch_dtype = fd
    df, tf = run(torch.float32)
    print('fine-branch', fd, 'backbone ms %.1f (fp32 %.1f)' % (tf * 1e3, t32 * 1e3))
    print('   coarse identical:', torch.equal(df['feats_c'], d32['feats_c']), ' ids identical:', torch.equal(df['i_ids'], d32['i_ids']) and torch.equal(df['j_ids'], d32['j_ids']))
    print('   featmap_f rel dev', ((df['featmap_f0'].float() - d32['featmap_f0']).norm() / d32['featmap_f0'].norm()).item())
    print('   mkpts1_f max abs dev px', (df['mkpts1_f'] - d32['mkpts1_f']).abs().max().item(), 'mean', (df['mkpts1_f'] - d32['mkpts1_f']).abs().mean().item())
m.backbone.fine_branch_dtype = None
d16, t16 = run(torch.float16)
print('backbone ms fp32 %.1f bf16 %.1f' % (t32 * 1e3, t16 * 1e3))
fc32, fc16 = d32['feats_c'], d16['feats_c']
print('feats_c rel dev', ((fc32 - fc16).norm() / fc32.norm()).item())
s32 = set(zip(d32['b_ids'].tolist(), d32['i_ids'].tolist(), d32['j_ids'].tolist()))
s16 = set(zip(d16['b_ids'].tolist(), d16['i_ids'].tolist(), d16['j_ids'].tolist()))
print('matches fp32 %d bf16 %d IoU %.4f' % (len(s32), len(s16), len(s32 & s16) / len(s32 | s16)))
