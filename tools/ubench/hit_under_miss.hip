// Development micro-benchmark: do L2 hits of one wave pass HBM misses of another wave of the same CU on the vector-memory path?
// One workgroup per CU (LDS request forces it), two waves: wave 0 streams a small L2-resident buffer (hits), wave 1 streams a
// buffer far larger than L2 + the die-level cache (misses); 8 x 16-byte loads in flight per lane.  Modes: hits alone, misses alone,
// both.  If the per-CU path returned strictly in order, the hit stream would slow to the miss stream's pace in mode "both".
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/hit_under_miss.hip -o /tmp/hum && /tmp/hum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(128) void k(const u32x4* hitbuf, const u32x4* missbuf, long miss_per_wg, int iters, int mode,
                                         unsigned long long* out, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((wave == 0 && !(mode & 1)) || (wave == 1 && !(mode & 2))) return;
    const u32x4* p = wave == 0 ? hitbuf + lane : missbuf + (long)blockIdx.x * miss_per_wg + lane;
    const long wrap = wave == 0 ? 4096 : miss_per_wg;          // hits: 64 KiB window (16-byte units), misses: the WG's own region
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    long off = 0;
    for (int i = 0; i < iters; ++i) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(p + off + j * 64) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(v[j])); acc += v[j].x ^ v[j].w; }
        off += 512;
        if (off + 512 > wrap) off = 0;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 2 + wave] = t1 - t0;
    if (acc == 0x12345u) sink[0] = acc;
}

int main() {
    const int nwg = 256, iters = 2000;
    const long miss_per_wg = 8L << 20 >> 4;                     // 8 MiB per workgroup in 16-byte units: 2 GiB in all
    u32x4 *hit, *miss; unsigned long long* out; unsigned* sink;
    hipMalloc(&hit, 1 << 20); hipMalloc(&miss, (size_t)nwg * miss_per_wg * 16); hipMalloc(&out, nwg * 2 * 8); hipMalloc(&sink, 4);
    hipMemset(hit, 1, 1 << 20); hipMemset(miss, 1, (size_t)nwg * miss_per_wg * 16);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int mode : {1, 2, 3, 1, 3}) {
        hipMemset(out, 0, nwg * 2 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(128), 100 * 1024, 0, hit, miss, miss_per_wg, iters, mode, out, sink);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("[kernel %.3f ms] ", ms);
        std::vector<unsigned long long> h(nwg * 2);
        hipMemcpy(h.data(), out, nwg * 2 * 8, hipMemcpyDeviceToHost);
        double s0 = 0, s1 = 0;
        for (int i = 0; i < nwg; ++i) { s0 += h[2 * i]; s1 += h[2 * i + 1]; }
        const double bytes = (double)iters * 8 * 1024;            // per wave
        printf("mode %d (%s): hit wave %.0f ticks (%.1f GB/s per CU at 100 MHz ticks), miss wave %.0f ticks (%.1f GB/s per CU)\n", mode,
               mode == 1 ? "hits alone" : mode == 2 ? "misses alone" : "both", s0 / nwg, s0 ? bytes / (s0 / nwg * 10e-9) / 1e9 : 0.0,
               s1 / nwg, s1 ? bytes / (s1 / nwg * 10e-9) / 1e9 : 0.0);
    }
    return 0;
}
