import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from far_amd import ops
from far_amd.loftr.transformer import LoFTREncoderLayer
torch.manual_seed(3)
layer = LoFTREncoderLayer(256, 8).cuda().eval()
g = torch.Generator(device='cuda').manual_seed(11)
N = int(os.environ.get('NIMG', '32'))
src = torch.randn(N, 4800, 256, device='cuda', generator=g)
with torch.no_grad():
    layer(src, src)
    pkv = [v for k, v in layer._packs._store.items() if isinstance(k, tuple) and k[0] == 'kv-state'][0]
    pkv = pkv[1] if isinstance(pkv, tuple) else pkv
    kv0, img0 = ops.linear_kv_state(src, pkv, 4800, want_image=True)
    kv0, img0 = kv0.clone(), img0.clone()
    nbad = 0
    for it in range(10):
        kv, img = ops.linear_kv_state(src, pkv, 4800, want_image=True)
        if not torch.equal(kv, kv0):
            nbad += 1
            d = (kv != kv0)
            if nbad <= 3:
                print('run', it, 'kv differs in', int(d.sum()), 'of', d.numel(), '| images', d.flatten(1).any(1).nonzero().flatten().tolist()[:40],
                      '| heads', sorted({int(r) // 32 for r in d.any(0).any(1).nonzero().flatten().tolist()}),
                      '| cols', d.any(0).any(0).nonzero().flatten().tolist(), '| max rel', float(((kv - kv0).abs() / (kv0.abs() + 1e-6)).max()))
    print('nondeterministic launches:', nbad, 'of 10; images', N)
