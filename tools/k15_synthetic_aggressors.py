"""Which synthetic kernel makes the REAL K15 (its source as it was: packed fp32) go wrong when it shares the CUs?  K15's source is
recompiled with the packed instructions on the GPU box and run on a side stream next to each aggressor of tools/ubench/dma_neighbour.hip
(and next to the real K13 as the control).  docs/rounds/r06.md section 2f.
python tools/k15_synthetic_aggressors.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ctypes.CDLL(os.path.join(ROOT, 'far_amd', 'lib', 'libfar_hip.so'), mode=ctypes.RTLD_GLOBAL)
import torch
from far_amd import ops, _lib
lib = _lib.load()
subprocess.check_call(['hipcc', '-shared', '-fPIC', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', os.path.join(ROOT, 'far_amd', 'csrc'),
                       os.path.join(ROOT, 'far_amd', 'csrc', 'head_linear_f32.hip'), '-o', '/tmp/k15_packed.so'])
subprocess.check_call(['hipcc', '-shared', '-fPIC', '--offload-arch=gfx950', '-O3', '-DAS_LIB', os.path.join(ROOT, 'tools', 'ubench', 'dma_neighbour.hip'),
                       '-o', '/tmp/libaggr.so'])
c_p, c_i, c_l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
v = ctypes.CDLL('/tmp/k15_packed.so')
v.far_rows_linear_f32.restype = c_i
v.far_rows_linear_f32.argtypes = [c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_p]
ag = ctypes.CDLL('/tmp/libaggr.so')
ag.launch_aggressor.argtypes = [c_i, c_p, c_p, c_p]
g = torch.Generator(device='cuda').manual_seed(78)
pr = ops.PackedRows(torch.randn(1024, 35840, device='cuda', generator=g) / 190)
feats = torch.randn(8, 35840, device='cuda', generator=g)
src = torch.full((1024 * 4096 + 8192,), 0x5a, dtype=torch.uint8, device='cuda')
sink = torch.zeros(16, device='cuda')
D = 128
pm = ops.PackedMlp(torch.randn(2 * D, 2 * D, device='cuda', generator=g) / 16, torch.randn(D, 2 * D, device='cuda', generator=g) / 16)
gam, bet = torch.rand(D, device='cuda', generator=g) + 0.5, torch.randn(D, device='cuda', generator=g) * 0.1
x = torch.randn(30000, 25, D, device='cuda', generator=g)
msg = torch.randn(30000, 25, D, device='cuda', generator=g)
side = torch.cuda.Stream()
def rows(stream):
    y = torch.empty(8, 1024, device='cuda')
    ws = torch.empty(lib.far_rows_linear_workspace_bytes(8, 1024, 35840), dtype=torch.uint8, device='cuda')
    assert v.far_rows_linear_f32(feats.data_ptr(), 35840, pr.packed.data_ptr(), None, None, 0, 8, 35840, 1024, 0, y.data_ptr(), 1024, ws.data_ptr(), stream.cuda_stream) == 0
    return y
ref = rows(torch.cuda.current_stream()); torch.cuda.synchronize(); ref = ref.clone()
names = ['LDS-DMA ring + ds_read_b128', 'global loads + ds_write + reads (no DMA)', 'MFMA only', 'MFMA only, 200 registers', 'VALU only, 256 registers',
         'LDS-DMA ring -> ds_read_b128 -> MFMA', 'ds_read_b128 -> MFMA (no DMA)', 'LDS-DMA ring -> ds_read_b128 (no MFMA)', 'fp16 split triples only',
         'fp16 split triples -> MFMA']
def aggr(mode):
    main = torch.cuda.current_stream()
    if mode < 0:
        ops.mlp_fused(x, msg, pm, gam, bet, 1e-5)
    else:
        assert ag.launch_aggressor(mode, src.data_ptr(), sink.data_ptr(), main.cuda_stream) == 0
for mode in [-1] + list(range(10)):
    bad = 0
    for it in range(10):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(2): aggr(mode)
        with torch.cuda.stream(side):
            y = rows(side)
        for _ in range(2): aggr(mode)
        ev1.record()
        torch.cuda.synchronize()
        bad += not torch.equal(y, ref)
    print(f'packed K15 next to [{"the real K13" if mode < 0 else names[mode]}]: {bad} of 10 launches differ   (aggressors: {ev0.elapsed_time(ev1):.1f} ms)')
