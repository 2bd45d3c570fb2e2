"""K2 (far_emm_pv_f16s, all passes) at the bench shape: 256 problems of 4800 x 4800 x 64, the shipped k_pv against far_set_tuning(14, 0 / 1) of a
library built with tools/experiments/k2_pv8_pipelined.patch (round 6; on the product library both settings run k_pv).  Usage: [FAR_HIP_LIB=...] python tools/k2_time.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from far_amd import _lib
lib = _lib.load()
g = torch.Generator(device='cuda').manual_seed(1)
Z, L = 256, 4800
q = torch.randn(Z, L, 64, device='cuda', generator=g); k = torch.randn(Z, L, 64, device='cuda', generator=g)
v = torch.randn(Z, L, 64, device='cuda', generator=g); pos = torch.rand(L, 6, device='cuda', generator=g)
T = torch.empty(Z, L, 70, device='cuda')
ws = torch.empty(lib.far_emm_pv_f16s_workspace_bytes(Z, L), dtype=torch.uint8, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    lib.far_emm_pv_f16s(q.data_ptr(), k.data_ptr(), v.data_ptr(), pos.data_ptr(), Z, L, 64, ctypes.c_float(0.125), 1, 0, L * 64, 0, ws.data_ptr(), T.data_ptr(), None, st)
ref = None
for v14 in (1, 0, 1, 0):                  # far_set_tuning(14, 1) = k_pv (rounds 3-5), 0 = k_pv8 (round 6); interleaved
    lib.far_set_tuning(14, v14)
    t = min(bench.event_time_ms(run, iters=3, warm=1) for _ in range(3))
    run()
    out = T.clone()
    if ref is None:
        ref = out
    print(f'far_emm_pv_f16s all passes, tuning 14={v14} ({"k_pv" if v14 else "k_pv8"}): {t:.3f} ms   bit-identical to the first run: {bool(torch.equal(out, ref))}   '
          f'checksum {float(out.double().sum()):.6e}', flush=True)
lib.far_set_tuning(14, 0)
