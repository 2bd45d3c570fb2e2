"""What K15's k_rows_partial needs to go wrong next to K13 on another stream (docs/rounds/r06.md section 2f): the kernel's source is
recompiled four ways on the GPU box -- as it was (packed fp32), without the packed instructions, with a static store loop (no
s_set_gpr_idx), with computed instead of streamed weights -- and each runs 15 times on a side stream next to the real K13.
Round 6: 14-15 / 0 / 15 / 0 of 15 launches differ.  (The shipped library is built without the packed instructions: far_amd/build.py.)
python tools/k15_victim_variants.py"""
import os, sys, ctypes, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ctypes.CDLL(os.path.join(ROOT, 'far_amd', 'lib', 'libfar_hip.so'), mode=ctypes.RTLD_GLOBAL)
import numpy as np, torch
from far_amd import ops, _lib
lib = _lib.load()
src = open(os.path.join(ROOT, 'far_amd', 'csrc', 'head_linear_f32.hip')).read()
STORE = "        for (int b = 0; b < B && b < RBT; ++b) partial[((size_t)s * B + b) * N + n] = acc[b];"
assert STORE in src
WLOAD = "                const float4 w = wp[(size_t)k4 * N + n];"
assert WLOAD in src
variants = {
    'A as shipped before (packed fp32)': (src, []),
    'B no packed fp32': (src, ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']),
    'C packed, static store loop (no s_set_gpr_idx)': (src.replace(STORE, "#pragma unroll\n        for (int b = 0; b < RBT; ++b) if (b < B) partial[((size_t)s * B + b) * N + n] = acc[b];"), []),
    'D packed, weights computed (no global loads in the loop)': (src.replace(WLOAD, "                const float4 w = make_float4(1e-3f * (n & 63) + 1e-4f * (k4 & 31), 0.5f, 0.25f - 1e-3f * (n & 7), 0.125f);"), []),
}
# E: the LDS tile written once in front of the loop (no barrier / ds_write inside it); F: no LDS reads in the loop (x from registers)
FILL = "        __syncthreads();\n        for (int i = threadIdx.x; i < RBT * KT4; i += 256) {"
assert FILL in src
XS = "                    const float4 v = xs[b][q];                      // same address in every lane: an LDS broadcast"
assert XS in src
variants['E packed, tile written once (no barrier in the loop)'] = (src.replace(FILL, "        if (sub == 0) __syncthreads();\n        for (int i = threadIdx.x; sub == 0 && i < RBT * KT4; i += 256) {"), [])
variants['F packed, x from registers (no LDS read in the loop)'] = (src.replace(XS, "                    const float4 v = make_float4(0.01f * b + 0.001f * q, 0.5f, 0.25f - 0.01f * b, 0.125f);"), [])
g = torch.Generator(device='cuda').manual_seed(78)
D, H = 128, 8
w0 = torch.randn(2 * D, 2 * D, device='cuda', generator=g) / 16
w2 = torch.randn(D, 2 * D, device='cuda', generator=g) / 16
gam, bet = torch.rand(D, device='cuda', generator=g) + 0.5, torch.randn(D, device='cuda', generator=g) * 0.1
pm = ops.PackedMlp(w0, w2)
n = 30000
x = torch.randn(n, 25, D, device='cuda', generator=g)
msg = torch.randn(n, 25, D, device='cuda', generator=g)
aggr = lambda: ops.mlp_fused(x, msg, pm, gam, bet, 1e-5)
W = torch.randn(1024, 35840, device='cuda', generator=g) / 190
pr = ops.PackedRows(W)
feats = torch.randn(8, 35840, device='cuda', generator=g)
side = torch.cuda.Stream()
c_p, c_i, c_l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
for name, (text, flags) in variants.items():
    tag = name.split()[0]
    open(f'/tmp/k15_{tag}.hip', 'w').write(text)
    r = subprocess.run(['hipcc', '-shared', '-fPIC', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', os.path.join(ROOT, 'far_amd', 'csrc'), *flags,
                        f'/tmp/k15_{tag}.hip', '-o', f'/tmp/k15_{tag}.so'], capture_output=True, text=True)
    if r.returncode:
        print(name, 'build failed', r.stderr[-300:]); continue
    v = ctypes.CDLL(f'/tmp/k15_{tag}.so')
    v.far_rows_linear_f32.restype = c_i
    v.far_rows_linear_f32.argtypes = [c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_p]
    def rows(stream):
        y = torch.empty(8, 1024, device='cuda')
        ws = torch.empty(lib.far_rows_linear_workspace_bytes(8, 1024, 35840), dtype=torch.uint8, device='cuda')
        rc = v.far_rows_linear_f32(feats.data_ptr(), 35840, pr.packed.data_ptr(), None, None, 0, 8, 35840, 1024, 0, y.data_ptr(), 1024, ws.data_ptr(), stream.cuda_stream)
        assert rc == 0
        return y
    ref = rows(torch.cuda.current_stream()); torch.cuda.synchronize(); ref = ref.clone()
    bad = 0
    for it in range(15):
        for _ in range(3): aggr()
        with torch.cuda.stream(side):
            y = rows(side)
        for _ in range(3): aggr()
        torch.cuda.synchronize()
        bad += not torch.equal(y, ref)
    print(f'K15 variant [{name}] next to K13: {bad} of 15 launches differ')
