// Development micro-benchmark: issue cost (cycles per wave64 instruction, s_memtime) of the VALU instructions K17's input transform is
// made of -- alone on a SIMD and next to a wave that issues MFMAs back to back on the same SIMD.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__device__ __forceinline__ unsigned long long run(float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 4; ++it) {
        if (OP == 0) asm volatile(REP64("v_fma_f32 %0, %1, %1, %0\n v_fma_f32 %2, %3, %3, %2\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (OP == 1) asm volatile(REP64("v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %2, %3, %3, %2\n") : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));
        if (OP == 2) asm volatile(REP64("v_pk_add_f32 %0, %1, %1\n v_pk_add_f32 %2, %3, %3\n") : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));
        if (OP == 3) asm volatile(REP64("v_cvt_pk_f16_f32 %0, %2, %3\n v_cvt_pk_f16_f32 %1, %3, %2\n") : "+v"(u0), "+v"(u1) : "v"(a0), "v"(a1));
        if (OP == 4) asm volatile(REP64("v_fma_mixlo_f16 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %1, %2, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n") : "+v"(u0), "+v"(u1) : "v"(u2), "v"(a1));
        if (OP == 5) asm volatile(REP64("v_cvt_f32_f16 %0, %2\n v_cvt_f32_f16 %1, %3\n") : "+v"(a0), "+v"(a1) : "v"(u0), "v"(u1));
        if (OP == 6) asm volatile(REP64("v_add_f32 %0, %1, %1\n v_sub_f32 %2, %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (OP == 7) asm volatile(REP64("v_cvt_f16_f32 %0, %2\n v_cvt_f16_f32 %1, %3\n") : "+v"(u0), "+v"(u1) : "v"(a0), "v"(a1));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + u0 + u1 + u2 + u3 == 12345.678f) printf("x");
    return t1 - t0;
}

__global__ __launch_bounds__(512) void k(unsigned long long* out, int with_mfma) {
    const int wave = threadIdx.x >> 6;
    if (wave >= 4) {                           // partner waves (same SIMDs as waves 0-3): MFMAs back to back, or idle
        if (!with_mfma) return;
        f32x16 acc = {0};
        f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
        for (int i = 0; i < 3000; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            f32x16 acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
            acc = acc2;
        }
        if (acc[0] == 123.f) out[100] = 1;
        return;
    }
    unsigned long long t[8];
    t[0] = run<0>(threadIdx.x); t[1] = run<1>(threadIdx.x); t[2] = run<2>(threadIdx.x); t[3] = run<3>(threadIdx.x);
    t[4] = run<4>(threadIdx.x); t[5] = run<5>(threadIdx.x); t[6] = run<6>(threadIdx.x); t[7] = run<7>(threadIdx.x);
    if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) out[i] = t[i];
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 1024);
    const char* names[8] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_cvt_pk_f16_f32", "v_fma_mixlo/hi_f16", "v_cvt_f32_f16", "v_add/sub_f32", "v_cvt_f16_f32"};
    for (int m = 0; m < 2; ++m) {
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, m);
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, m);
        unsigned long long h[8];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s:\n", m ? "next to a wave issuing MFMAs back to back" : "alone on the SIMD");
        for (int i = 0; i < 8; ++i) printf("  %-22s %.2f cycles per instruction\n", names[i], (double)h[i] / (4 * 128));
    }
    return 0;
}
