import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from far_amd import ops
import far_amd.ops.linear as OL
from far_amd.loftr.transformer import LoFTREncoderLayer
torch.manual_seed(3)
layer = LoFTREncoderLayer(256, 8).cuda().eval()
g = torch.Generator(device='cuda').manual_seed(11)
x = torch.randn(32, 4800, 256, device='cuda', generator=g)
src = torch.randn(32, 4800, 256, device='cuda', generator=g)
REC = []
def wrap(mod, name):
    f = getattr(mod, name)
    def w(*a, **k):
        r = f(*a, **k)
        outs = r if isinstance(r, (tuple, list)) else (r,)
        REC.append((name, [t.clone() for t in outs if torch.is_tensor(t)]))
        return r
    setattr(mod, name, w); setattr(ops, name, w)
for n in ('linear_f16s', 'linear_kv_state', 'linear_q_apply'):
    wrap(OL, n)
with torch.no_grad():
    layer(x, src); torch.cuda.synchronize()
    first = list(REC)
    bad = {}
    for it in range(20):
        del REC[:]
        layer(x, src); torch.cuda.synchronize()
        for i, ((n, a), (_, b)) in enumerate(zip(first, REC)):
            for j, (p, q) in enumerate(zip(a, b)):
                if not torch.equal(p, q):
                    d = (p.float() - q.float()).abs()
                    bad.setdefault((i, n, j), []).append((int((d > 0).sum()), float(d.max())))
print('calls per layer:', [(i, n, [tuple(t.shape) for t in a]) for i, (n, a) in enumerate(first)])
for k, v in sorted(bad.items()):
    print('NONDETERMINISTIC', k, 'in', len(v), 'of 20 runs; e.g. (n elements, max diff)', v[:3])
print('done', len(bad))
