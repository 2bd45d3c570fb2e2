import sys; sys.path.insert(0,'.')
import torch
from far_amd import ops, _lib
lib=_lib.load()
def timeit_rot(fns, n=3):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n*len(fns))
def run(rows,Cin,Cout,planes=1,nbuf=12):
    xs=[torch.randn(1,1,rows,Cin,device='cuda') for _ in range(nbuf)]; w=torch.randn(Cout,Cin,device='cuda')*0.05
    pc=ops.PackedConv(w)
    fl=2.0*rows*Cin*Cout
    out=[]
    for mode in (0,1):
        lib.far_set_tuning(3,mode)
        t=timeit_rot([ (lambda x=x: ops.conv_nhwc(x,pc,out_planes=planes)) for x in xs])
        t1=timeit_rot([ (lambda x=xs[0]: ops.conv_nhwc(x,pc,out_planes=planes)) ]*4)
        out.append(f'{"gemm" if mode==0 else "conv1"}: cold {t*1000:.0f} us hot {t1*1000:.0f} us')
    lib.far_set_tuning(3,0)
    print(f'rows={rows} {Cin}->{Cout} planes={planes}: '+' | '.join(out)+f'  (ideal@300TF {fl/300e9*1e3:.0f} us)',flush=True)
run(153600,256,256); run(153600,256,768,3); run(153600,256,512,2); run(153600,512,512); run(153600,512,256)
run(1500000,128,128,1,4); run(1500000,128,384,3,4); run(1500000,256,256,1,4); run(1500000,256,128,1,4)
