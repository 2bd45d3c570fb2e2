// K1, bf16-input variant (see the banner below).  Shares the similarity parameters, the selection / compaction
// kernels and the workspace layout with the exact-fp32 variant (dual_softmax_common.h).
#include "dual_softmax_common.h"

using namespace far_ds;

namespace {

// ============================================================================================
// bf16-input variant of K1 (north_star: "MFMA bf16 where they are genuine dense contractions").
// Features are rounded once to bf16 (RNE); the contraction runs on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation; everything after the dot product (scaling, softmax statistics, P, selection) is the fp32 code of
// the exact variant.  Parity statement: bit-exact indices / 1e-5 confidences against the oracle evaluated on the
// SAME bf16-rounded features; against the fp32 path it is reported as match-set IoU (tests/test_coarse_gpu.py).
// The row panel (32 rows x C channels per wave) is register resident (C/4 VGPRs); the other map streams through a
// double-buffered ring of 64-column half tiles filled by LDS-DMA (global_load_lds_dwordx4), XOR-swizzled on the
// source side so the ds_read_b128 fragment reads are conflict free; one barrier per half tile.
// ============================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua += 0x7fffu + ((ua >> 16) & 1u);          // round to nearest even
    ub += 0x7fffu + ((ub >> 16) & 1u);
    return (ua >> 16) | (ub & 0xffff0000u);
}

__global__ void k_cvt_bf16(const float4* __restrict__ in, uint2* __restrict__ out, long nvec) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const float4 v = in[i];
        out[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
}

__device__ __forceinline__ void softmax_merge2(float& m, float& s, float mo, float so) {   // log2-domain partials
    const float mn = fmaxf(m, mo);
    s = s * __builtin_amdgcn_exp2f(m - mn) + so * __builtin_amdgcn_exp2f(mo - mn);
    m = mn;
}

__global__ void k_colreduce2(const float2* __restrict__ colpart, int nI, int S, float2* __restrict__ colstat) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, z = blockIdx.y;
    if (j >= S) return;
    float m = NEG_BIG, s = 0.f;
    for (int b = 0; b < nI; ++b) {
        float2 v = colpart[((size_t)z * nI + b) * S + j];
        softmax_merge2(m, s, v.x, v.y);
    }
    colstat[(size_t)z * S + j] = make_float2(m, s);
}

typedef __attribute__((address_space(1))) const void* far_gptr_t;
typedef __attribute__((address_space(3))) void* far_lptr_t;

template <int NS>   // NS = C / 16 MFMA k-steps (C = 256 -> 16)
struct Bf16Tile {
    static constexpr int C = NS * 16;
    static constexpr int ROWB = C * 2;               // bytes per LDS row (unpadded: the LDS-DMA image is lane-linear)
    static constexpr int SLOTS = ROWB / 16;          // 16-byte slots per row (32)
    static constexpr int HALF_BYTES = 64 * ROWB;     // one 64-column half tile (32 KiB)
    uint4 afr[NS];                                   // A fragments: row i, channels 16 s + 8 h .. + 7

    __device__ __forceinline__ void load_a(const uint16_t* __restrict__ A, int irow, int L, int h) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            afr[s] = irow < L ? *reinterpret_cast<const uint4*>(A + (size_t)irow * C + 16 * s + 8 * h) : make_uint4(0, 0, 0, 0);
    }
    // Asynchronous global -> LDS copy (LDS-DMA, global_load_lds_dwordx4) of 64 rows [jrow0, jrow0+64) of B into `dst`.
    // One wave-instruction moves 1 KiB = 2 rows; the destination is lane-linear, so the bank-conflict swizzle
    // (16-byte slot ^= row & 15) is applied to the per-lane SOURCE address and again on the read side
    // (cdna_hip_programming.md rule 21).  Rows past S are clamped (their columns are masked in the epilogue).
    __device__ __forceinline__ static void stage_half_async(char* dst, const uint16_t* __restrict__ B, int jrow0, int S,
                                                            int wave, int lane) {
        static_assert(SLOTS == 32, "bf16 tile engine is written for C = 256");
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pair = wave * 8 + q;
            const int R = 2 * pair + (lane >> 5);
            const int g = (lane & 31) ^ (R & 15);
            const int jr = min(jrow0 + R, S - 1);
            const char* src = reinterpret_cast<const char*>(B) + (size_t)jr * ROWB + g * 16;
            __builtin_amdgcn_global_load_lds((far_gptr_t)src, (far_lptr_t)(dst + pair * 1024), 16, 0, 0);
        }
    }
    // acc[ct] = A (32 x C) . B[32 ct + (0..31)]^T for ct = 0, 1 of the half tile at `src`   (D[m = i][n = j])
    __device__ __forceinline__ void mma_half(f32x16 (&acc)[2], const char* src, int l31, int h) const {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, afr[s]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int row = 32 * ct + l31;
                const uint4 bu = *reinterpret_cast<const uint4*>(src + row * ROWB + (((2 * s + h) ^ (row & 15)) * 16));
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bu), acc[ct], 0, 0, 0);
            }
        }
    }
};

// Statistics of one 32 x 64 half tile held as two accumulators (lane: column 32 ct + l31, rows mfma32_row(r, h)).
// Works in the log2 domain (one v_exp_f32 per term, no extra multiply).  FULL = no padding, no masks: no selects.
template <bool FULL>
__device__ __forceinline__ void stats_half_epilogue(f32x16 (&acc)[2], const SimParams& sp, float (&rm)[16], float (&rs)[16],
                                                    float (&cmx)[2], float (&csm)[2], unsigned rowvalid, unsigned rowmasked,
                                                    const bool* cvalid, const bool* cmasked) {
    const float fill2 = sp.mask_fill * 1.44269504088896341f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = acc[ct][r] * sp.k2;
            if (!FULL) { if (cmasked[ct] || ((rowmasked >> r) & 1)) s = fill2; }
            acc[ct][r] = s;
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float tm;
        if (FULL) tm = fmaxf(acc[0][r], acc[1][r]);
        else {
            tm = NEG_BIG;
            if (cvalid[0]) tm = fmaxf(tm, acc[0][r]);
            if (cvalid[1]) tm = fmaxf(tm, acc[1][r]);
        }
        const float mn = fmaxf(rm[r], tm);
        float sum = rs[r] * __builtin_amdgcn_exp2f(rm[r] - mn);       // rm = -FLT_MAX initially: 2^-inf = 0, rs = 0
        if (FULL) sum += __builtin_amdgcn_exp2f(acc[0][r] - mn) + __builtin_amdgcn_exp2f(acc[1][r] - mn);
        else {
            if (cvalid[0]) sum += __builtin_amdgcn_exp2f(acc[0][r] - mn);
            if (cvalid[1]) sum += __builtin_amdgcn_exp2f(acc[1][r] - mn);
        }
        rm[r] = mn;
        rs[r] = sum;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        float m = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (FULL || ((rowvalid >> r) & 1)) m = fmaxf(m, acc[ct][r]);
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (FULL || ((rowvalid >> r) & 1)) s += __builtin_amdgcn_exp2f(acc[ct][r] - m);
        const float mo = shfl_xor_f(m, 32), so = shfl_xor_f(s, 32);
        const float mn = fmaxf(m, mo);
        s = s * __builtin_amdgcn_exp2f(m - mn) + so * __builtin_amdgcn_exp2f(mo - mn);
        cmx[ct] = mn;
        csm[ct] = s;
    }
}

template <int NS>
__global__ __launch_bounds__(256, 2) void k_stats_bf16(
    const uint16_t* __restrict__ f0, const uint16_t* __restrict__ f1, int Z, int L, int S, SimParams sp,
    const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
    float2* __restrict__ rowstat, float2* __restrict__ colpart) {
    typedef Bf16Tile<NS> T;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* bt = smem_raw;                                                                        // [2][64][C] bf16
    float2* colx = reinterpret_cast<float2*>(smem_raw + 2 * T::HALF_BYTES);                     // [2][4][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nI = (L + TILE_M - 1) / TILE_M;
    int z, Ib;
    tile_coords(nI, Z, z, Ib);
    const int i0 = Ib * TILE_M;
    const uint16_t* A = f0 + (size_t)z * L * T::C;
    const uint16_t* B = f1 + (size_t)z * S * T::C;
    T tile;
    tile.load_a(A, i0 + 32 * wave + l31, L, h);
    unsigned rowvalid = 0, rowmasked = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + 32 * wave + mfma32_row(r, h);
        if (i < L) {
            rowvalid |= 1u << r;
            if (mask0 && !mask0[(size_t)z * L + i]) rowmasked |= 1u << r;
        }
    }
    float rm[16], rs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { rm[r] = NEG_BIG; rs[r] = 0.f; }
    // software pipeline over 64-column half tiles: the LDS-DMA of half t+1 is in flight while half t is computed;
    // one barrier per half (it also publishes the column partials of that half)
    const int nH = (S + 63) / 64;
    T::stage_half_async(bt, B, 0, S, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
#pragma unroll 1
        for (int t = 0; t < nH; ++t) {
            const int jh = 64 * t;                       // first column of this half
            const char* cur = bt + (t & 1) * T::HALF_BYTES;
            if (t + 1 < nH) T::stage_half_async(bt + ((t + 1) & 1) * T::HALF_BYTES, B, jh + 64, S, wave, lane);
            float2* colh = colx + (t & 1) * 256;
            f32x16 acc[2];
            tile.mma_half(acc, cur, l31, h);
            const int j0 = jh, jb = 0;
            float cmx[2], csm[2];
            // log2-domain scores: s2 = dot * (acc_scale / temperature * log2 e); statistics are (max2, sum 2^(s2-max2))
            // must be WAVE-UNIFORM: the epilogue exchanges values across the lanes of the wave (half swap)
            const bool full = (jh + 64 <= S) && (i0 + 32 * wave + 32 <= L) && !mask0 && !mask1;
            if (full) {
                stats_half_epilogue<true>(acc, sp, rm, rs, cmx, csm, 0xffffu, 0u, nullptr, nullptr);
            } else {
                bool cvalid[2], cmasked[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int j = j0 + jb + 32 * ct + l31;
                    cvalid[ct] = j < S;
                    cmasked[ct] = cvalid[ct] && mask1 && !mask1[(size_t)z * S + j];
                }
                stats_half_epilogue<false>(acc, sp, rm, rs, cmx, csm, rowvalid, rowmasked, cvalid, cmasked);
            }
            if (h == 0) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) colh[wave * 64 + 32 * ct + l31] = make_float2(cmx[ct], csm[ct]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA pieces of half t+1 have landed
            __syncthreads();
            if (tid < 64 && j0 + jb + tid < S) {
                float2 v = colh[tid];
                float m = v.x, s = v.y;
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float2 o = colh[w * 64 + tid];
                    softmax_merge2(m, s, o.x, o.y);
                }
                colpart[((size_t)z * nI + Ib) * S + j0 + jb + tid] = make_float2(m, s);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float m = rm[r], s = rs[r];
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const float mo = shfl_xor_f(m, d), so = shfl_xor_f(s, d);
            softmax_merge2(m, s, mo, so);
        }
        if (l31 == 0 && ((rowvalid >> r) & 1))
            rowstat[(size_t)z * L + i0 + 32 * wave + mfma32_row(r, h)] = make_float2(m, s);
    }
}

template <int NS>
__global__ __launch_bounds__(256, 2) void k_match_bf16(
    const uint16_t* __restrict__ f0, const uint16_t* __restrict__ f1, int Z, int L, int S, SimParams sp,
    const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
    const float2* __restrict__ rowstat, const float2* __restrict__ colstat, float* __restrict__ conf,
    float* __restrict__ rowbest_v, int* __restrict__ rowbest_j, float* __restrict__ colbest_part) {
    typedef Bf16Tile<NS> T;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* bt = smem_raw;                                                                        // [2][64][C] bf16
    float* colx = reinterpret_cast<float*>(smem_raw + 2 * T::HALF_BYTES);                       // [2][4][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nI = (L + TILE_M - 1) / TILE_M;
    int z, Ib;
    tile_coords(nI, Z, z, Ib);
    const int i0 = Ib * TILE_M;
    const uint16_t* A = f0 + (size_t)z * L * T::C;
    const uint16_t* B = f1 + (size_t)z * S * T::C;
    T tile;
    tile.load_a(A, i0 + 32 * wave + l31, L, h);
    unsigned rowvalid = 0, rowmasked = 0;
    float rmax[16], rinv[16], bestv[16];
    int bestj[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + 32 * wave + mfma32_row(r, h);
        rmax[r] = 0.f; rinv[r] = 1.f; bestv[r] = -1.f; bestj[r] = 0x7fffffff;
        if (i < L) {
            rowvalid |= 1u << r;
            if (mask0 && !mask0[(size_t)z * L + i]) rowmasked |= 1u << r;
            const float2 st = rowstat[(size_t)z * L + i];
            rmax[r] = st.x; rinv[r] = 1.0f / st.y;
        }
    }
    const int nH = (S + 63) / 64;
    T::stage_half_async(bt, B, 0, S, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
#pragma unroll 1
        for (int t = 0; t < nH; ++t) {
            const int j0 = 64 * t, jb = 0;
            const char* cur = bt + (t & 1) * T::HALF_BYTES;
            if (t + 1 < nH) T::stage_half_async(bt + ((t + 1) & 1) * T::HALF_BYTES, B, j0 + 64, S, wave, lane);
            float* colh = colx + (t & 1) * 256;
            f32x16 acc[2];
            tile.mma_half(acc, cur, l31, h);
            float cbest[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int j = j0 + jb + 32 * ct + l31;
                const bool cvalid = j < S;
                const bool cmasked = cvalid && mask1 && !mask1[(size_t)z * S + j];
                const float2 cst = cvalid ? colstat[(size_t)z * S + j] : make_float2(0.f, 1.f);
                const float cinv = 1.0f / cst.y;
                float cb = -1.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = acc[ct][r] * sp.k2;                                   // log2 domain (statistics too)
                    if (cmasked || ((rowmasked >> r) & 1)) s = sp.mask_fill * 1.44269504088896341f;
                    const float p = (__builtin_amdgcn_exp2f(s - cst.x) * cinv) * (__builtin_amdgcn_exp2f(s - rmax[r]) * rinv[r]);
                    if (cvalid && ((rowvalid >> r) & 1)) {
                        if (conf) conf[((size_t)z * L + i0 + 32 * wave + mfma32_row(r, h)) * S + j] = p;
                        if (p > bestv[r]) { bestv[r] = p; bestj[r] = j; }
                        cb = fmaxf(cb, p);
                    }
                }
                cbest[ct] = fmaxf(cb, shfl_xor_f(cb, 32));
            }
            if (h == 0) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) colh[wave * 64 + 32 * ct + l31] = cbest[ct];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < 64 && j0 + jb + tid < S)
                colbest_part[((size_t)z * nI + Ib) * S + j0 + jb + tid] =
                    fmaxf(fmaxf(colh[tid], colh[64 + tid]), fmaxf(colh[128 + tid], colh[192 + tid]));
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = bestv[r];
        int j = bestj[r];
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const float vo = shfl_xor_f(v, d);
            const int jo = shfl_xor_i(j, d);
            if (vo > v || (vo == v && jo < j)) { v = vo; j = jo; }
        }
        if (l31 == 0 && ((rowvalid >> r) & 1)) {
            const int i = i0 + 32 * wave + mfma32_row(r, h);
            rowbest_v[(size_t)z * L + i] = v;
            rowbest_j[(size_t)z * L + i] = j;
        }
    }
}

}  // namespace

extern "C" {

size_t far_coarse_match_bf16_workspace_bytes(int Z, int L, int S, int C) {
    return carve(nullptr, Z, L, S).bytes + align256((size_t)Z * L * C * 2) + align256((size_t)Z * S * C * 2);
}

// bf16-input variant of far_coarse_match_f32 (same arguments and outputs; C must be 256).
int far_coarse_match_bf16(const float* f0, const float* f1, int Z, int L, int S, int C,
                          float temperature, float thr, int border, int h0, int w0, int h1, int w1,
                          float cell_scale, const uint8_t* mask0, const uint8_t* mask1,
                          const int* valid_hw, const float* scale0, const float* scale1,
                          float* conf_out, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                          float* mkpts0_c, float* mkpts1_c, int* counts_out, int* total_out,
                          void* ws, hipStream_t stream) {
    far_clear_errors();
    if (!f0 || !f1 || !ws || !b_ids || !i_ids || !j_ids || !mconf || !mkpts0_c || !mkpts1_c || !total_out)
        return FAR_EINVAL;
    if (Z <= 0 || L <= 0 || S <= 0 || C != 256 || h0 * w0 != L || h1 * w1 != S) return FAR_EINVAL;
    K1Workspace w = carve(ws, Z, L, S);
    uint16_t* f0b = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(ws) + w.bytes);
    uint16_t* f1b = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(f0b) + align256((size_t)Z * L * C * 2));
    SimParams sp = make_sim(sqrtf((float)C), temperature, 1.0f);
    if (sp.feat_div != 1.0f) return FAR_EINVAL;   // sqrt(C) must be a power of two (C = 256, 64) for the folded scaling
    const long n0 = (long)Z * L * C / 4, n1 = (long)Z * S * C / 4;
    hipLaunchKernelGGL(k_cvt_bf16, dim3(2048), dim3(256), 0, stream, (const float4*)f0, (uint2*)f0b, n0);
    hipLaunchKernelGGL(k_cvt_bf16, dim3(2048), dim3(256), 0, stream, (const float4*)f1, (uint2*)f1b, n1);
    const int nI = (L + TILE_M - 1) / TILE_M;
    int* counts = counts_out ? counts_out : w.counts;
    hipMemsetAsync(counts, 0, sizeof(int) * Z, stream);
    const size_t smem = 2 * 64 * (size_t)C * 2 + 2 * 4 * 64 * sizeof(float2);
    FAR_ONCE_PER_DEVICE(
        hipFuncSetAttribute((const void*)k_stats_bf16<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute((const void*)k_match_bf16<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
#define FAR_BF16_LAUNCH(NS)                                                                                          \
    hipLaunchKernelGGL(k_stats_bf16<NS>, dim3(nI * Z), dim3(256), smem, stream, f0b, f1b, Z, L, S, sp, mask0, mask1, \
                       w.rowstat, w.colpart);                                                                        \
    hipLaunchKernelGGL(k_colreduce2, dim3((S + 255) / 256, Z), dim3(256), 0, stream, w.colpart, nI, S, w.colstat);   \
    hipLaunchKernelGGL(k_match_bf16<NS>, dim3(nI * Z), dim3(256), smem, stream, f0b, f1b, Z, L, S, sp, mask0, mask1, \
                       w.rowstat, w.colstat, conf_out, w.rowbest_v, w.rowbest_j, w.colbest_part);
    FAR_BF16_LAUNCH(16)
#undef FAR_BF16_LAUNCH
    hipLaunchKernelGGL(k_finalize, dim3((L + 255) / 256, Z), dim3(256), 0, stream, w.rowbest_v, w.rowbest_j,
                       w.colbest_part, nI, L, S, thr, border, h0, w0, h1, w1, valid_hw, w.match_j, counts);
    hipLaunchKernelGGL(k_compact, dim3(Z), dim3(256), 0, stream, w.match_j, w.rowbest_v, counts, L, w0, w1,
                       cell_scale, scale0, scale1, b_ids, i_ids, j_ids, mconf, mkpts0_c, mkpts1_c, total_out);
    return far_check_launch();
}

}  // extern "C"
