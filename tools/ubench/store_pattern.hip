// Micro-benchmark: HBM write bandwidth of the conf_matrix store patterns (no arithmetic).
//   0  linear fill, dwordx4 per lane (what torch.fill_ does)
//   1  k1_conf_wide pattern: WG = 4 waves side by side (64 columns each), 32-row steps; per wave-instruction 2 rows x 128 B (dword)
//   2  same tile walk, but each wave-instruction writes 1 row x 1 KiB (dwordx4 per lane, 4 instructions cover 4 rows... ) i.e.
//      WG step = 32 rows x 256 columns written as 8 instructions per wave of (4 rows x 256 B)?  -> variant: lanes 0-15 = 256 B of a row
//   3  pattern 1 with dwordx2 (lane pairs)  [not used]
// build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int Lr = 4800, Sc = 4800;

__global__ __launch_bounds__(256) void fill_linear(float4* p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(1, 2, 3, 4);
}

// grid: Z * nJ * nch items as in k1_conf_wide (nJ = 19, 10 chunks of 15 row tiles)
__global__ __launch_bounds__(256, 2) void tile_dword(float* conf, int Z, int mode) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nJ = 19, nch = 10, tpc = 15;
    const int id = blockIdx.x;
    const int xcd = id & 7, li = id >> 3;
    const int z = xcd + 8 * (li / (nJ * nch));
    const int item = li % (nJ * nch);
    const int Jb = item / nch, ch = item % nch;
    const int col0 = Jb * 256 + 64 * wave;
    if (col0 >= Sc) return;
    float* base = conf + (size_t)z * Lr * Sc;
    for (int it = ch * tpc; it < (ch + 1) * tpc; ++it) {
        if (mode == 1) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int col = col0 + 32 * ct + l31;
                    if (col < Sc) base[(size_t)row * Sc + col] = (float)r;
                }
        } else if (mode == 2) {
            // 16 lanes x float4 = 256 B of one row; the wave covers 4 rows per instruction, 8 instructions = 32 rows
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = it * 32 + 4 * k + (lane >> 4);
                const int col = col0 + 4 * (lane & 15);
                if (col < Sc) *reinterpret_cast<float4*>(base + (size_t)row * Sc + col) = make_float4(1, 2, 3, 4);
            }
        } else if (mode == 3) {
            // whole WG writes 1 row x 1 KiB per instruction: lane -> 16 B, wave -> 1 KiB contiguous of one row; 8 rows per wave
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = it * 32 + 8 * wave + k;
                const int col = Jb * 256 + 4 * lane;
                if (col < Sc) *reinterpret_cast<float4*>(base + (size_t)row * Sc + col) = make_float4(1, 2, 3, 4);
            }
        }
    }
}

int main() {
    const int Z = 32;
    const size_t n = (size_t)Z * Lr * Sc;
    float* d;
    hipMalloc(&d, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(fill_linear, dim3(256 * 16), dim3(256), 0, 0, (float4*)d, n / 4);
            else hipLaunchKernelGGL(tile_dword, dim3(Z * 19 * 10), dim3(256), 0, 0, d, Z, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("mode %d: %.3f ms  %.0f GB/s\n", mode, best, n * 4 / best / 1e6);
    }
    return 0;
}
