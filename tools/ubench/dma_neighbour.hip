// Does a kernel that streams into LDS with global_load_lds (LDS-DMA) disturb a DIFFERENT kernel that shares its CU?
//
// Round 6 found K15 (head_linear_f32.hip: k_rows_partial -- 32 accumulators in v0..v31, broadcast ds_read_b128 of a 16 KiB static
// LDS tile) computing wrong partial sums whenever it ran on a second stream next to K13 / K14 (asm global_load_lds_dwordx4 rings):
// always lanes 48..63, always the even accumulators, 30 launches of 30; never next to K17 (same asm, but one workgroup fills the CU),
// never next to ATen kernels, never alone (docs/rounds/r06.md section 2f).  This program isolates the pair:
//
//   victim    256 threads, 16 KiB static LDS filled once with a known pattern, 32 integer accumulators per thread fed by broadcast
//             ds_read_b128 -- K15's inner loop with exact arithmetic; at the end it re-checks its LDS tile against the pattern.
//             Mode R: the same accumulation from registers only (no LDS reads in the loop).
//   aggressor 256 threads, 48 KiB dynamic LDS, every wave requests 1 KiB pieces into LDS in a loop and reads them back:
//             mode 0 = asm global_load_lds_dwordx4 with the M0 save / restore of far_amd's kernels
//             mode 1 = the same without restoring M0            mode 2 = __builtin_amdgcn_global_load_lds, 16 bytes per lane
//             mode 3 = __builtin_amdgcn_global_load_lds, 4 bytes per lane      mode 4 = global_load_dwordx4 + ds_write_b128 (no DMA)
//
// The victim runs alone first (reference), then next to each aggressor on a second stream.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_neighbour.hip -o /tmp/dma_nb && /tmp/dma_nb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(1))) const void* gptr_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned pat(int b, int q, int c) { return 1u + 7u * b + 131u * q + 17u * c; }

template <bool REGS>
__global__ __launch_bounds__(256) void victim(unsigned* __restrict__ out, unsigned* __restrict__ lds_err, int iters) {
    __shared__ u32x4 xs[32][32];
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int b = i >> 5, q = i & 31;
        xs[b][q] = u32x4{pat(b, q, 0), pat(b, q, 1), pat(b, q, 2), pat(b, q, 3)};
    }
    __syncthreads();
    unsigned acc[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) acc[b] = 0;
    const unsigned w = 3u + 2u * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
        for (int q = 0; q < 32; ++q) {
            const unsigned wq = w + (unsigned)(q + it);
#pragma unroll
            for (int b = 0; b < 32; ++b) {
                u32x4 v;
                if (REGS) v = u32x4{pat(b, q, 0), pat(b, q, 1), pat(b, q, 2), pat(b, q, 3)};
                else v = xs[b][q];
                acc[b] = acc[b] * 3u + v.x * wq;
                acc[b] = acc[b] * 5u + v.y * wq;
                acc[b] = acc[b] * 7u + v.z * wq;
                acc[b] = acc[b] * 9u + v.w * wq;
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 32; ++b) out[((size_t)blockIdx.x * 32 + b) * 256 + threadIdx.x] = acc[b];
    __syncthreads();
    unsigned bad = 0;
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int b = i >> 5, q = i & 31;
        const u32x4 v = xs[b][q];
        bad += v.x != pat(b, q, 0) || v.y != pat(b, q, 1) || v.z != pat(b, q, 2) || v.w != pat(b, q, 3);
    }
    if (bad) atomicAdd(lds_err, bad);
}

// K15's inner loop as it is: fp32 fma chains, which hipcc packs into v_pk_fma_f32 (two accumulators per instruction, op_sel broadcasts).
// Build with -DNO_PK (adds nothing here; pass  -Xclang -target-feature -Xclang -packed-fp32-ops  to hipcc) for the unpacked control.
__global__ __launch_bounds__(256) void victim_f32(float* __restrict__ out, int iters, const float4* __restrict__ wp) {
    __shared__ float4 xs[32][32];
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int b = i >> 5, q = i & 31;
        xs[b][q] = make_float4(0.25f + 0.001f * b + 0.01f * q, 0.5f - 0.002f * b, 0.125f + 0.003f * q, 1.0f / (1 + b + q));
    }
    __syncthreads();
    float acc[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) acc[b] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
        for (int q = 0; q < 32; ++q) {
            float4 w;
            if (wp) {
                w = wp[(size_t)((blockIdx.x * 5 + it * 32 + q) & 4095) * 256 + threadIdx.x];      // K15's weight stream: 1 KiB per wave and k-step
            } else {
                const float f = 1.0f / (1 + ((threadIdx.x * 7 + q * 3 + it) & 255));
                w = make_float4(f, 0.5f * f, 0.25f * f, -0.125f * f);
            }
#pragma unroll
            for (int b = 0; b < 32; ++b) {
                const float4 v = xs[b][q];
                acc[b] = __builtin_fmaf(v.x, w.x, acc[b]);
                acc[b] = __builtin_fmaf(v.y, w.y, acc[b]);
                acc[b] = __builtin_fmaf(v.z, w.z, acc[b]);
                acc[b] = __builtin_fmaf(v.w, w.w, acc[b]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 32; ++b) out[((size_t)blockIdx.x * 32 + b) * 256 + threadIdx.x] = acc[b];
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MFMA only: no memory traffic at all inside the loop
__global__ __launch_bounds__(256) void aggressor_mfma(float* __restrict__ sink, int iters) {
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x & 15) + 0.001f * i); b[i] = (_Float16)(0.02f * (threadIdx.x & 7) - 0.001f * i); }
    f32x16 c0, c1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += c0[i] + c1[i];
    if (t == 1234.5f) sink[0] = t;
}

// MFMA only, with a register footprint like K13 / K14's (12 accumulator tiles = 192 registers + operands): a co-resident victim wave
// is then allocated in the upper half of the SIMD's 512-entry register file
__global__ __launch_bounds__(256) void aggressor_mfma_big(float* __restrict__ sink, int iters) {
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x & 15) + 0.001f * i); b[i] = (_Float16)(0.02f * (threadIdx.x & 7) - 0.001f * i); }
    f32x16 c[12];
#pragma unroll
    for (int j = 0; j < 12; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) t += c[j][i];
    if (t == 1234.5f) sink[0] = t;
}
// VALU only, the same footprint: 200 live registers, no matrix instruction
__global__ __launch_bounds__(256) void aggressor_valu_big(float* __restrict__ sink, int iters) {
    float r[200];
#pragma unroll
    for (int i = 0; i < 200; ++i) r[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 200; ++i) r[i] = __builtin_fmaf(r[i], 1.0001f, 0.5f);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 200; ++i) t += r[i];
    if (t == 1234.5f) sink[0] = t;
}

// Half-register writes only: the fp16 split's v_cvt_pk_f16_f32 + v_fma_mixlo / mixhi_f16 triple in a loop (what K9 / K13 / K14 issue
// thousands of times per tile next to their MFMAs), optionally with MFMAs consuming the results
template <bool WITH_MFMA>
__global__ __launch_bounds__(256) void aggressor_mix(float* __restrict__ sink, int iters) {
    float x0 = 0.001f * threadIdx.x, x1 = 0.5f + 0.002f * threadIdx.x;
    unsigned acc = 0;
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned lo4[4], hi4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h, l;
            asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(h), "=&v"(l) : "v"(x0), "v"(x1));
            hi4[j] = h; lo4[j] = l;
            x0 += 0.25f; x1 -= 0.125f;
        }
        if (WITH_MFMA) {
            const u32x4 a = u32x4{hi4[0], hi4[1], hi4[2], hi4[3]}, b = u32x4{lo4[0], lo4[1], lo4[2], lo4[3]};
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        } else {
            acc += hi4[0] ^ lo4[1] ^ hi4[2] ^ lo4[3];
        }
    }
    float t = (float)acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += c[i];
    if (t == 1234.5f) sink[0] = t;
}

__device__ __forceinline__ void glds16_restore(const void* gsrc, unsigned lds_dst_uniform);
// The shape of K9 / K13 / K14: LDS-DMA ring pieces, fragments read back with ds_read_b128, fed to MFMAs (VARIANT 0); 1 = without the
// DMA (the LDS tile is written once with ds_write); 2 = without the MFMAs (reads xor-ed); 3 = MFMAs on register operands + the DMA + reads unused
template <int VARIANT>
__global__ __launch_bounds__(256) void aggressor_gemm(const unsigned char* __restrict__ src, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)lds);
    f32x16 c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    f16x8 ra, rb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ra[i] = (_Float16)(0.01f * (lane & 15)); rb[i] = (_Float16)(0.02f * (lane & 7)); }
    if (VARIANT == 1) {
        for (int j = 0; j < 12; ++j)
            *reinterpret_cast<u32x4*>(lds + (unsigned)(wave * 12 + j) * 1024 + lane * 16) = u32x4{0x3c003c00u, 0x38003800u, 0x34003400u, 0x30003000u};
        __syncthreads();
    }
    unsigned x = 0;
    for (int it = 0; it < iters; ++it) {
        if (VARIANT != 1) {
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const unsigned off = (unsigned)(wave * 12 + j) * 1024;
                const unsigned char* g = src + ((size_t)((it * 7 + blockIdx.x * 3 + j) & 1023) * 4096) + off % 4096;
                glds16_restore(g + lane * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(base + off)));
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < 12; j += 2) {
            const f16x8 a = *reinterpret_cast<const f16x8*>(lds + (unsigned)(((wave + 1) & 3) * 12 + j) * 1024 + lane * 16);
            const f16x8 b = *reinterpret_cast<const f16x8*>(lds + (unsigned)(((wave + 1) & 3) * 12 + j + 1) * 1024 + lane * 16);
            if (VARIANT == 2) {
                const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
                x += ua.x ^ ua.y ^ ub.z ^ ub.w;
            } else if (VARIANT == 3) {
                const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
                x += ua.x ^ ub.w;
                c[(j >> 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra, rb, c[(j >> 1) & 3], 0, 0, 0);
                c[((j >> 1) + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb, ra, c[((j >> 1) + 1) & 3], 0, 0, 0);
            } else {
                c[(j >> 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[(j >> 1) & 3], 0, 0, 0);
                c[((j >> 1) + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c[((j >> 1) + 1) & 3], 0, 0, 0);
            }
        }
        if (VARIANT != 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float t = (float)x;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) t += c[j][i];
    if (t == 1234.5f) sink[0] = t;
}

__device__ __forceinline__ void glds16_restore(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ void glds16_norestore(const void* gsrc, unsigned lds_dst_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void aggressor(const unsigned char* __restrict__ src, unsigned* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)lds);
    unsigned s = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {                       // 12 KiB per wave and round: 48 KiB per workgroup
            const unsigned off = (unsigned)(wave * 12 + j) * 1024;
            const unsigned char* g = src + ((size_t)((it * 7 + blockIdx.x * 3 + j) & 1023) * 4096) + off % 4096;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + off));
            if (MODE == 0) glds16_restore(g + lane * 16, dst);
            else if (MODE == 1) glds16_norestore(g + lane * 16, dst);
            else if (MODE == 2) __builtin_amdgcn_global_load_lds((gptr_t)(g + lane * 16), (lptr_t)(lds + off), 16, 0, 0);
            else if (MODE == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __builtin_amdgcn_global_load_lds((gptr_t)(g + k * 256 + lane * 4), (lptr_t)(lds + off + k * 256), 4, 0, 0);
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(g + lane * 16);
                *reinterpret_cast<u32x4*>(lds + off + lane * 16) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + (unsigned)(((wave + 1) & 3) * 12 + j) * 1024 + lane * 16);
            s += v.x ^ v.y ^ v.z ^ v.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (s == 0x12345678u) sink[0] = s;
}

template <bool REGS>
static void run_victim(hipStream_t st, unsigned* out, unsigned* err, int blocks, int iters) {
    hipLaunchKernelGGL(victim<REGS>, dim3(blocks), dim3(256), 0, st, out, err, iters);
}

// -DAS_LIB: the fp32 victim as a shared library for tools/k15_synthetic_victim.py (next to the REAL K13 of libfar_hip.so)
extern "C" int launch_victim_f32(float* out, int iters, const void* wp, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(victim_f32, dim3(blocks), dim3(256), 0, st, out, iters, (const float4*)wp);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ... and the synthetic aggressors for tools/k15_synthetic_aggressors.py (next to the REAL packed K15)
extern "C" int launch_aggressor(int mode, const void* src, void* sink, hipStream_t st) {
    static bool once = false;
    if (!once) {
        once = true;
        hipFuncSetAttribute((const void*)aggressor<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
        hipFuncSetAttribute((const void*)aggressor<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
        hipFuncSetAttribute((const void*)aggressor_gemm<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
        hipFuncSetAttribute((const void*)aggressor_gemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
        hipFuncSetAttribute((const void*)aggressor_gemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    }
    const unsigned char* s8 = (const unsigned char*)src;
    switch (mode) {
        case 0: hipLaunchKernelGGL(aggressor<0>, dim3(2048), dim3(256), 49152, st, s8, (unsigned*)sink, 600); break;             // LDS-DMA ring + reads
        case 1: hipLaunchKernelGGL(aggressor<4>, dim3(2048), dim3(256), 49152, st, s8, (unsigned*)sink, 600); break;             // loads + ds_write + reads
        case 2: hipLaunchKernelGGL(aggressor_mfma, dim3(4096), dim3(256), 0, st, (float*)sink, 3000); break;                    // MFMA only
        case 3: hipLaunchKernelGGL(aggressor_mfma_big, dim3(4096), dim3(256), 0, st, (float*)sink, 1200); break;                // MFMA, 200 registers
        case 4: hipLaunchKernelGGL(aggressor_valu_big, dim3(4096), dim3(256), 0, st, (float*)sink, 200); break;                 // VALU, 256 registers
        case 5: hipLaunchKernelGGL(aggressor_gemm<0>, dim3(2048), dim3(256), 49152, st, s8, (float*)sink, 600); break;          // DMA ring -> reads -> MFMA
        case 6: hipLaunchKernelGGL(aggressor_gemm<1>, dim3(2048), dim3(256), 49152, st, s8, (float*)sink, 1600); break;         // reads -> MFMA, no DMA
        case 7: hipLaunchKernelGGL(aggressor_gemm<2>, dim3(2048), dim3(256), 49152, st, s8, (float*)sink, 600); break;          // DMA ring -> reads, no MFMA
        case 8: hipLaunchKernelGGL(aggressor_mix<false>, dim3(4096), dim3(256), 0, st, (float*)sink, 4000); break;              // fp16 split triples
        case 9: hipLaunchKernelGGL(aggressor_mix<true>, dim3(4096), dim3(256), 0, st, (float*)sink, 2000); break;               // split triples -> MFMA
        default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

#ifndef AS_LIB
int main() {
    const int VB = 4096, AB = 8192, VIT = 40, AIT = 60;
    unsigned *out, *ref, *err, *sink;
    unsigned char* src;
    const size_t n = (size_t)VB * 32 * 256;
    CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&ref, n * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&src, 1024 * 4096 + 8192));
    CK(hipMemset(src, 0x5a, 1024 * 4096 + 8192));
    hipStream_t sa, sv;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sv));
    CK(hipFuncSetAttribute((const void*)aggressor<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    CK(hipFuncSetAttribute((const void*)aggressor<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    CK(hipFuncSetAttribute((const void*)aggressor<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    CK(hipFuncSetAttribute((const void*)aggressor<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    CK(hipFuncSetAttribute((const void*)aggressor<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    std::vector<unsigned> h(n), hr(n);
    const char* names[5] = {"asm dwordx4, M0 saved / restored", "asm dwordx4, M0 left", "builtin, 16 bytes per lane", "builtin, 4 bytes per lane",
                            "no DMA (global load + ds_write)"};
    for (int regs = 0; regs < 2; ++regs) {
        CK(hipMemset(err, 0, 4));
        if (regs) run_victim<true>(sv, ref, err, VB, VIT); else run_victim<false>(sv, ref, err, VB, VIT);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hr.data(), ref, n * 4, hipMemcpyDeviceToHost));
        for (int mode = 0; mode < 5; ++mode) {
            CK(hipMemset(err, 0, 4));
            CK(hipMemset(out, 0, n * 4));
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1, a0, a1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
            CK(hipEventRecord(a0, sa));
            switch (mode) {
                case 0: hipLaunchKernelGGL(aggressor<0>, dim3(AB), dim3(256), 49152, sa, src, sink, AIT); break;
                case 1: hipLaunchKernelGGL(aggressor<1>, dim3(AB), dim3(256), 49152, sa, src, sink, AIT); break;
                case 2: hipLaunchKernelGGL(aggressor<2>, dim3(AB), dim3(256), 49152, sa, src, sink, AIT); break;
                case 3: hipLaunchKernelGGL(aggressor<3>, dim3(AB), dim3(256), 49152, sa, src, sink, AIT); break;
                default: hipLaunchKernelGGL(aggressor<4>, dim3(AB), dim3(256), 49152, sa, src, sink, AIT); break;
            }
            CK(hipEventRecord(a1, sa));
            CK(hipEventRecord(e0, sv));
            if (regs) run_victim<true>(sv, out, err, VB, VIT); else run_victim<false>(sv, out, err, VB, VIT);
            CK(hipEventRecord(e1, sv));
            CK(hipDeviceSynchronize());
            float vms, ams;
            CK(hipEventElapsedTime(&vms, e0, e1)); CK(hipEventElapsedTime(&ams, a0, a1));
            CK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost));
            unsigned lerr;
            CK(hipMemcpy(&lerr, err, 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            size_t by_q[4] = {0, 0, 0, 0}, by_par[2] = {0, 0};
            for (size_t i = 0; i < n; ++i)
                if (h[i] != hr[i]) {
                    ++bad;
                    ++by_q[(i & 63) >> 4];
                    ++by_par[(i >> 8) & 1];
                }
            printf("victim (%s) next to aggressor [%s]: %zu of %zu accumulators wrong (lane groups 0-15 / 16-31 / 32-47 / 48-63: %zu %zu %zu %zu; even / odd "
                   "accumulator: %zu %zu), LDS tile words found changed at the end: %u   [victim %.2f ms, aggressor %.2f ms]\n",
                   regs ? "registers only" : "broadcast ds_read_b128", names[mode], bad, n, by_q[0], by_q[1], by_q[2], by_q[3], by_par[0], by_par[1], lerr,
                   vms, ams);
        }
    }
    // ---- the fp32 victim (v_pk_fma_f32) next to: nothing, the LDS-DMA aggressor, the MFMA-only aggressor
    {
        float *fo, *fr, *fsink;
        CK(hipMalloc(&fo, n * 4)); CK(hipMalloc(&fr, n * 4)); CK(hipMalloc(&fsink, 4));
        std::vector<float> hf(n), hfr(n);
        float4* wts;
        CK(hipMalloc(&wts, (size_t)4096 * 256 * 16));
        {
            std::vector<float> hw((size_t)4096 * 256 * 4);
            for (size_t i = 0; i < hw.size(); ++i) hw[i] = 1.0f / (1 + (i * 2654435761u >> 24));
            CK(hipMemcpy(wts, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        }
        const float4* wsel = getenv("NO_WLOAD") ? nullptr : wts;
        printf("fp32 victim: weights %s\n", wsel ? "streamed from global memory (as K15)" : "computed");
        hipLaunchKernelGGL(victim_f32, dim3(VB), dim3(256), 0, sv, fr, VIT, wsel);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hfr.data(), fr, n * 4, hipMemcpyDeviceToHost));
        CK(hipFuncSetAttribute((const void*)aggressor_gemm<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
        CK(hipFuncSetAttribute((const void*)aggressor_gemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
        CK(hipFuncSetAttribute((const void*)aggressor_gemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
        CK(hipFuncSetAttribute((const void*)aggressor_gemm<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
        const char* an[11] = {"nothing", "asm dwordx4 LDS-DMA", "MFMA only (no memory traffic)", "LDS-DMA ring -> ds_read_b128 -> MFMA (K9 / K13 / K14's shape)",
                             "ds_read_b128 -> MFMA, no DMA", "LDS-DMA ring -> ds_read_b128, no MFMA", "LDS-DMA ring + reads + MFMA on register operands",
                             "MFMA only, 200+ registers per wave", "VALU only, 200+ registers per wave",
                              "fp16 split triples (half-register writes) only", "fp16 split triples feeding MFMAs"};
        for (int mode = 0; mode < 11; ++mode)
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(fo, 0, n * 4));
                CK(hipDeviceSynchronize());
                if (mode == 1) hipLaunchKernelGGL(aggressor<0>, dim3(AB), dim3(256), 49152, sa, src, sink, AIT);
                if (mode == 2) hipLaunchKernelGGL(aggressor_mfma, dim3(16384), dim3(256), 0, sa, fsink, 4000);
                if (mode == 3) hipLaunchKernelGGL(aggressor_gemm<0>, dim3(AB), dim3(256), 49152, sa, src, fsink, AIT);
                if (mode == 4) hipLaunchKernelGGL(aggressor_gemm<1>, dim3(AB), dim3(256), 49152, sa, src, fsink, 4 * AIT);
                if (mode == 5) hipLaunchKernelGGL(aggressor_gemm<2>, dim3(AB), dim3(256), 49152, sa, src, fsink, AIT);
                if (mode == 9) hipLaunchKernelGGL(aggressor_mix<false>, dim3(16384), dim3(256), 0, sa, fsink, 6000);
                if (mode == 10) hipLaunchKernelGGL(aggressor_mix<true>, dim3(16384), dim3(256), 0, sa, fsink, 3000);
                if (mode == 7) hipLaunchKernelGGL(aggressor_mfma_big, dim3(16384), dim3(256), 0, sa, fsink, 1500);
                if (mode == 8) hipLaunchKernelGGL(aggressor_valu_big, dim3(16384), dim3(256), 0, sa, fsink, 300);
                if (mode == 6) hipLaunchKernelGGL(aggressor_gemm<3>, dim3(AB), dim3(256), 49152, sa, src, fsink, AIT);
                hipLaunchKernelGGL(victim_f32, dim3(VB), dim3(256), 0, sv, fo, VIT, wsel);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hf.data(), fo, n * 4, hipMemcpyDeviceToHost));
                size_t bad = 0, by_q[4] = {0, 0, 0, 0}, by_par[2] = {0, 0};
                double worst = 0;
                for (size_t i = 0; i < n; ++i)
                    if (hf[i] != hfr[i]) {
                        ++bad; ++by_q[(i & 63) >> 4]; ++by_par[(i >> 8) & 1];
                        const double d = fabs((double)hf[i] - hfr[i]);
                        worst = d > worst ? d : worst;
                    }
                printf("fp32 victim next to [%s]: %zu of %zu accumulators differ from the solo run (lane groups: %zu %zu %zu %zu; even / odd accumulator: %zu %zu; "
                       "max |diff| %.3e)\n", an[mode], bad, n, by_q[0], by_q[1], by_q[2], by_q[3], by_par[0], by_par[1], worst);
            }
    }
    return 0;
}
#endif
