"""Event-timed K2 (far_emm_pv_f16s, all passes) at the step's shape: 32 pairs x 8 problems of 4800 x 4800."""
import ctypes, sys
sys.path.insert(0, '.')
import torch
import bench
from far_amd import _lib          # FAR_HIP_LIB=far_amd/lib/libfar_hip_base.so (tools/ab_build.py) times another build
lib = _lib.load()
Z, L = 256, 4800
g = torch.Generator(device='cuda').manual_seed(1)
q = torch.randn(Z, L, 64, device='cuda', generator=g); k = torch.randn(Z, L, 64, device='cuda', generator=g); v = torch.randn(Z, L, 64, device='cuda', generator=g)
pos = torch.rand(L, 6, device='cuda', generator=g); T = torch.empty(Z, L, 70, device='cuda')
ws = torch.empty(lib.far_emm_pv_f16s_workspace_bytes(Z, L), dtype=torch.uint8, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    lib.far_emm_pv_f16s(q.data_ptr(), k.data_ptr(), v.data_ptr(), pos.data_ptr(), Z, L, 64, ctypes.c_float(0.125), 1, 0, L * 64, 0, ws.data_ptr(), T.data_ptr(), None, st)
print('far_emm_pv_f16s all passes: %.3f ms' % bench.event_time_ms(run, iters=5, warm=2))
