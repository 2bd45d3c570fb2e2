"""Is tools/ubench/dma_neighbour.hip's synthetic victim (K15's loop: v_pk_fma_f32 chains, broadcast ds_read_b128, streamed weights)
vulnerable at all?  It is built as a library on the GPU box and run on a side stream next to the REAL K13 of libfar_hip.so -- the
aggressor that made the real K15 go wrong 30 times of 30 (docs/rounds/r06.md section 2f).
python tools/k15_synthetic_victim.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from far_amd import ops
subprocess.check_call(['hipcc', '-shared', '-fPIC', '--offload-arch=gfx950', '-O3', '-DAS_LIB', os.path.join(ROOT, 'tools', 'ubench', 'dma_neighbour.hip'),
                       '-o', '/tmp/libvictim.so'])
v = ctypes.CDLL('/tmp/libvictim.so')
v.launch_victim_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
g = torch.Generator(device='cuda').manual_seed(78)
D = 128
pm = ops.PackedMlp(torch.randn(2 * D, 2 * D, device='cuda', generator=g) / 16, torch.randn(D, 2 * D, device='cuda', generator=g) / 16)
gam, bet = torch.rand(D, device='cuda', generator=g) + 0.5, torch.randn(D, device='cuda', generator=g) * 0.1
x = torch.randn(30000, 25, D, device='cuda', generator=g)
msg = torch.randn(30000, 25, D, device='cuda', generator=g)
aggr = lambda: ops.mlp_fused(x, msg, pm, gam, bet, 1e-5)
VB, IT = 1024, 10
wts = torch.rand(4096 * 256 * 4, device='cuda', generator=g) + 0.01
side = torch.cuda.Stream()
def victim(stream, streamed):
    out = torch.empty(VB * 32 * 256, device='cuda')
    assert v.launch_victim_f32(out.data_ptr(), IT, wts.data_ptr() if streamed else None, VB, stream.cuda_stream) == 0
    return out
for streamed in (True, False):
    ref = victim(torch.cuda.current_stream(), streamed); torch.cuda.synchronize(); ref = ref.clone()
    bad = 0
    for it in range(15):
        for _ in range(3): aggr()
        with torch.cuda.stream(side):
            y = victim(side, streamed)
        for _ in range(3): aggr()
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1
            if bad == 1:
                d = (y != ref).nonzero().flatten()
                print('   first differing launch:', len(d), 'accumulators; lane groups', sorted({(int(i) & 63) >> 4 for i in d[:4000].tolist()}),
                      'accumulator parity', sorted({(int(i) >> 8) & 1 for i in d[:4000].tolist()}))
    print(f'synthetic victim ({"streamed" if streamed else "computed"} weights) next to the real K13: {bad} of 15 launches differ')
