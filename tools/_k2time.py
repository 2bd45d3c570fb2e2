import sys; sys.path.insert(0,'.')
import torch
from far_amd import ops
from tools.conv_probe import timeit
Z,N=256,4800
q=torch.randn(Z,N,64,device='cuda'); k=torch.randn(Z,N,64,device='cuda'); v=torch.randn(Z,N,64,device='cuda'); pos=torch.rand(N,6,device='cuda')
for ex in (True, False):
    t=timeit(lambda: ops.emm_bilinear(q,k,v,pos,0.125,exact_f32=ex), n=3)
    print('exact_f32' if ex else 'f16s', f'{t:.2f} ms (incl. bmm/cat)')
