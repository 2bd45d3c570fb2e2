import sys; sys.path.insert(0,'.')
import torch
from far_amd import ops
from tools.conv_probe import timeit
n=32; L=4800
g=torch.Generator(device='cuda').manual_seed(1)
f0=1.2*torch.randn(n,L,256,device='cuda',generator=g)
f1=f0[:,torch.randperm(L,device='cuda',generator=g)]+0.1*torch.randn(n,L,256,device='cuda',generator=g)
for v in ('f32','bf16','f16s'):
    t=timeit(lambda: ops.coarse_match(f0,f1,0.1,0.2,2,(60,80),(60,80),8.0,variant=v), n=3)
    tc=timeit(lambda: ops.coarse_match(f0[:16],f1[:16],0.1,0.2,2,(60,80),(60,80),8.0,variant=v,want_conf=True), n=3)
    by=(4.0*L*L+4.0*2*L*256)*16
    print(f'{v}: all passes {t:.2f} ms; materialising (16 pairs) {tc:.2f} ms = {by/tc/1e6:.0f} GB/s')
