import sys; sys.path.insert(0,'.')
import torch
from far_amd import ops, _lib
from tools.conv_probe import timeit
lib=_lib.load()
def run(N,H,W,Cin,Cout,ks,split,cfg=0,dbg=0):
    lib.far_set_tuning(3,cfg); lib.far_set_tuning(2,dbg)
    x=torch.randn(N,H,W,Cin,device='cuda').relu_(); w=torch.randn(Cout,Cin,ks,ks,device='cuda')*0.05
    res=torch.randn(N,H,W,Cout,device='cuda')
    pc=ops.PackedConv(w,torch.ones(Cout,device='cuda'),torch.zeros(Cout,device='cuda'),split=split)
    fl=2.0*N*H*W*Cin*Cout*ks*ks
    t=timeit(lambda: ops.conv_nhwc(x,pc,residual=res,act='relu'))
    t2=timeit(lambda: ops.conv_nhwc(x,pc,residual=None,act='relu'))
    print(f'{Cin}->{Cout} k{ks} {H}x{W} split={split} cfg={cfg} dbg={dbg}: {t:.3f} ms ({fl/t/1e9:.0f} TF/s)  nores {t2:.3f}',flush=True)
    lib.far_set_tuning(3,0); lib.far_set_tuning(2,0)
for dbg in (0,1):
  for cfg in (0,1):
    run(64,240,320,128,128,3,True,cfg,dbg)
    run(64,120,160,256,256,3,True,cfg,dbg)
    run(64,120,160,196,196,3,True,cfg,dbg)
    run(64,120,160,256,256,3,False,cfg,dbg)
    run(1,1,307200,512,512,1,True,cfg,dbg)
