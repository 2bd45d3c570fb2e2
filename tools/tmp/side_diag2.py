import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from far_amd import synth, ops
from far_amd.config import far_eval_config
from far_amd.loftr import LoFTR
m = LoFTR(far_eval_config()).eval(); synth.load_synthetic(m, seed=0); m = m.cuda()
m.head_prefetch = False
im0, im1 = synth.synth_image_pair(8, seed=21)
d = {'image0': torch.from_numpy(im0).cuda(), 'image1': torch.from_numpy(im1).cuda()}
with torch.no_grad():
    m(d)
    torch.cuda.synchronize()
    t0, t1 = d['featmap0'], d['featmap1']
    head = m.loftr_regress
    def cf():
        with ops.activation_exponent(m.act_exp):
            f = head.compute_features(t0, t1, None, None)
        return f
    a = cf(); torch.cuda.synchronize()
    b = cf(); torch.cuda.synchronize()
    print('main vs main', torch.equal(a.feats, b.feats), torch.equal(a.enc0, b.enc0))
    side = torch.cuda.Stream()
    for trial in range(3):
        with torch.cuda.stream(side):
            c = cf()
        torch.cuda.synchronize()
        print('main vs side (idle main)', torch.equal(a.feats, c.feats), torch.equal(a.enc0, c.enc0), float((a.enc0 - c.enc0).abs().max()))
    # with a busy main stream
    x = torch.randn(8192, 8192, device='cuda')
    for trial in range(3):
        with torch.cuda.stream(side):
            c = cf()
        for _ in range(20):
            y = x @ x
        torch.cuda.synchronize()
        print('main vs side (busy main)', torch.equal(a.feats, c.feats), torch.equal(a.enc0, c.enc0), float((a.enc0 - c.enc0).abs().max()))
    # rows_linear alone on the side stream
    pk = head._packs
    both = pk.get('enc0|moe0', [head.encoder[0].weight, head.moe_predictor[0].weight], lambda: 1/0)
    r0 = ops.rows_linear(a.feats, both); torch.cuda.synchronize()
    for trial in range(3):
        with torch.cuda.stream(side):
            r1 = ops.rows_linear(a.feats, both)
        torch.cuda.synchronize()
        print('rows_linear main vs side', torch.equal(r0, r1), float((r0 - r1).abs().max()))
