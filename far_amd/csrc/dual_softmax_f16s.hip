// K1 on the f16 matrix cores with split-precision operands: all-pairs correlation + dual softmax + mutual-NN.
//
// Same operator and outputs as far_coarse_match_f32 (dual_softmax_f32.hip; reference
// mp3d_loftr/src/loftr/utils/coarse_matching.py:86-265), built like K2's split variant (emm_bilinear_f16s.hip):
//   k1_prep      feat -> fp16 hi / lo planes, scaled by 2^4, stored in the LDS image ([rows][256] fp16, 16-byte
//                slot ^= row & 15) so that a 64-row tile is one linear LDS-DMA
//   k1_rowstats  x2: log2-domain softmax statistics of every row of the similarity matrix -- (f0, f1) for the softmax
//                over columns j, (f1, f0) for the softmax over rows i (a lane owns one row of the transposed score
//                tile: no cross-lane reductions); the second call also leaves dense, padded column arrays
//   k1_screen    (round 4, unmasked non-materialising calls) hi.hi scores + a norm bound -> which 32-row x 64-column tiles can hold
//                an entry with conf > thr at all; k1_match skips the others (3 % survive on the bench workload)
//   k1_match     conf = 2^(2 x - rowmax - colmax) / (rowsum colsum): ONE exp per score; per row the best (conf, j)
//                (ties -> smaller j, as mask.max(dim=2) on CPU), per 128-row block the column maxima (32-lane max of
//                each accumulator register); optional conf_matrix with 16-byte stores
// then k_finalize / k_compact of dual_softmax_common.h (threshold, border, mutual-NN, ordered compaction).
// The contraction is three v_mfma_f32_32x32x16_f16 per 16 channels (hi.hi + hi.lo + lo.hi, fp32 accumulate): an
// fp32-grade similarity at 16/3 of the exact-f32 MFMA rate of dual_softmax_f32.hip.
#include "dual_softmax_common.h"
#include <algorithm>

// dual_softmax_conf_f16.hip: the HBM-bound conf_matrix writer (plain-fp16 scores + exact fix-up of the non-tiny entries)
int far_k1_conf_launch(const _Float16* ah, const _Float16* bh, int Z, int L, int S, int Lp, int Sp, float c1, float fill2,
                       const uint8_t* mask0, const uint8_t* mask1, const float2* rowstat, const float* cmax,
                       const float* cinv, float* conf, const int* fix_count, const uint2* fix_list, int slots,
                       int* fix_info_out, hipStream_t stream);

namespace {

using namespace far_ds;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int C = 256;               // channels (the FAR coarse level)
constexpr int NS = C / 16;           // MFMA k-steps
constexpr int KT = 64;               // columns per tile
constexpr int ROWB = C * 2;          // bytes per fp16 row
constexpr float PRESCALE = 16.0f;
constexpr float HUGE_F = 1.0e30f;
constexpr int TILE_PLANE = KT * ROWB;   // 32 KiB
constexpr int CAND_SLOTS = 4;           // exact-entry slots per (column, half-wave) of the conf_matrix writer

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// max over the 32 lanes of each half-wave; the result is valid in lanes 16..31 (and 48..63) of the wave.
#define FAR_DPP(old, v, ctrl, rowmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), ctrl, rowmask, 0xF, false))
__device__ __forceinline__ float max32_to_upper_row(float v) {
    v = fmaxf(v, FAR_DPP(v, v, 0xB1, 0xF));      // quad_perm [1,0,3,2]
    v = fmaxf(v, FAR_DPP(v, v, 0x4E, 0xF));      // quad_perm [2,3,0,1]
    v = fmaxf(v, FAR_DPP(v, v, 0x141, 0xF));     // row_half_mirror
    v = fmaxf(v, FAR_DPP(v, v, 0x140, 0xF));     // row_mirror: every lane of a 16-lane row holds the row maximum
    v = fmaxf(v, FAR_DPP(v, v, 0x142, 0xA));     // row_bcast:15 into rows 1 and 3: lanes 16-31 / 48-63 hold the 32-lane maximum
    return v;
}

// x [Z][N][256] fp32 -> hi / lo [Z][Np][256] fp16 (rows >= N zero), slot ^= row & 15
// overflow (device int or null): |= 1 when a feature is beyond the range of the 2^4-scaled split (|x| > 4094: hi = inf)
// nmax (device uint or null): atomic maximum of the bit pattern of max_rows sum_c (2^4 x)^2 -- the squared norm bound of the match
// pass's tile prescreen (non-negative floats order like unsigned integers); the caller zeroes it.
__global__ void k1_prep(const float* __restrict__ x, int Z, int N, int Np, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                        int* __restrict__ overflow, unsigned* __restrict__ nmax) {
    const long total = (long)Z * Np * 32;
    bool bad = false;
    float n2max = 0.f;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int slot = (int)(t & 31);
        const long row = t >> 5;
        const unsigned urow = (unsigned)row, uz = urow / (unsigned)Np;      // Z Np < 2^31 (host check): 32-bit division
        const int i = (int)(urow - uz * (unsigned)Np);
        const long z = uz;
        f16x8 vh, vl;
        if (i < N) {
            const float* src = x + ((size_t)z * N + i) * C + slot * 8;
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 h, l;
                split1(v[e] * PRESCALE, h, l);
                vh[e] = h; vl[e] = l;
                bad |= !(fabsf(v[e]) <= 65504.0f / PRESCALE);          // also true for NaN inputs
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { vh[e] = (_Float16)0.f; vl[e] = (_Float16)0.f; }
        }
        const int s2 = slot ^ (i & 15);
        *reinterpret_cast<f16x8*>(hi + (size_t)row * C + s2 * 8) = vh;
        *reinterpret_cast<f16x8*>(lo + (size_t)row * C + s2 * 8) = vl;
        if (nmax) {                                  // the row's 32 slots are the 32 lanes of a half-wave (total is a multiple of 32)
            float n2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float q = (float)vh[e] + (float)vl[e]; n2 = fmaf(q, q, n2); }
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) n2 += shfl_xor_f(n2, m);
            n2max = fmaxf(n2max, n2);
        }
    }
    if (nmax) {
        n2max = fmaxf(n2max, shfl_xor_f(n2max, 32));
        // one atomic per wave on one address would serialise 78 k of them in L2 (0.8 ms): only a wave that can still raise the
        // maximum issues it (the plain read may be stale -- then the atomic is merely redundant; the maximum only grows)
        if ((threadIdx.x & 63) == 0 && n2max > __uint_as_float(*reinterpret_cast<volatile unsigned*>(nmax))) atomicMax(nmax, __float_as_uint(n2max));
    }
    if (overflow && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(overflow, 1);
}

// Row-side fragments: row i, channels 16 s + 8 h .. + 7, both planes (128 registers)
struct RowFrags {
    f16x8 hi[NS], lo[NS];
    __device__ __forceinline__ void load(const _Float16* ph, const _Float16* pl, size_t row, int irow, int h) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int slot = (2 * s + h) ^ (irow & 15);
            hi[s] = *reinterpret_cast<const f16x8*>(ph + row * C + 8 * slot);
            lo[s] = *reinterpret_cast<const f16x8*>(pl + row * C + 8 * slot);
        }
    }
};

__device__ __forceinline__ void dma_tile(unsigned char* lds, const _Float16* gh, const _Float16* gl, size_t row0, int tid, int wave) {
    const unsigned char* sh = reinterpret_cast<const unsigned char*>(gh + row0 * C) + tid * 16;
    const unsigned char* sl = reinterpret_cast<const unsigned char*>(gl + row0 * C) + tid * 16;
#pragma unroll
    for (int j = 0; j < TILE_PLANE / 4096; ++j) {
        __builtin_amdgcn_global_load_lds((gptr_t)(sh + j * 4096), (lptr_t)(lds + j * 4096 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(sl + j * 4096), (lptr_t)(lds + TILE_PLANE + j * 4096 + wave * 1024), 16, 0, 0);
    }
}

// acc[ct]: D[m = tile row 32 ct + ..][n = this lane's row]
__device__ __forceinline__ void score_tile(f32x16 (&acc)[2], const unsigned char* lds, const RowFrags& rf, int l31, int h) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f16x8 ch[2], cl[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int row = 32 * ct + l31;
            const int off = row * ROWB + (((2 * s + h) ^ (row & 15)) * 16);
            ch[ct] = *reinterpret_cast<const f16x8*>(lds + off);
            cl[ct] = *reinterpret_cast<const f16x8*>(lds + TILE_PLANE + off);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[ct], rf.hi[s], acc[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[ct], rf.lo[s], acc[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl[ct], rf.hi[s], acc[ct], 0, 0, 0);
    }
}

// stat[z][i] = (max_j x_ij, sum_j 2^(x_ij - max)) over the Nc real columns; masked pairs count with x = fill2
// (masked_fill_(-INF), coarse_matching.py:108-111).  Optional dense padded copies dmax / dinv [Z][Nrp] (+huge / 0 past Nr).
// CAND (second call of the conf_matrix writer, rows = the matrix's COLUMNS j, tile columns = its rows i whose statistics
// are final): every entry that can reach conf >= 2^-12 (row term exact, column term bounded from above by the running
// column statistics) is appended to this lane's PRIVATE slot list cand[(z Nr + j) * 2 + h][CAND_SLOTS] as (i, bits of x)
// -- no atomics, no cross-lane traffic: four slots per half-column, the count (which may exceed the slots: overflow is
// reported, never silent) in cand_count -- and the writer's fix-up evaluates those
// from THIS x, the very value the statistics were accumulated from (an independently computed "exact" x would not
// cancel against them: the fp32 accumulation error of a 256-term dot product is ~1e-5 in the log2 domain).
template <bool CAND>
__global__ __launch_bounds__(256, 2) void k1_rowstats(const _Float16* __restrict__ ah, const _Float16* __restrict__ al,
                                                      const _Float16* __restrict__ bh, const _Float16* __restrict__ bl,
                                                      int Z, int Nr, int Nc, int Nrp, int Ncp, float c1, float fill2,
                                                      const uint8_t* __restrict__ rmask, const uint8_t* __restrict__ cmask,
                                                      float2* __restrict__ stat, float* __restrict__ dmax, float* __restrict__ dinv,
                                                      float* __restrict__ dthr, const float* __restrict__ othr,
                                                      int* __restrict__ cand_count, uint2* __restrict__ cand) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    int z, Ib;
    tile_coords(Nrp / 128, Z, z, Ib);
    const int irow = Ib * 128 + 32 * wave + l31;
    RowFrags rf;
    rf.load(ah, al, (size_t)z * Nrp + irow, irow, h);
    const bool rmasked = rmask && irow < Nr && !rmask[(size_t)z * Nr + irow];
    float m = -HUGE_F, sum = 0.f, comp = 0.f;
    int ncand = 0;
    const int ntile = (Nc + KT - 1) / KT;
    for (int jt = 0; jt < ntile; ++jt) {
        __syncthreads();
        dma_tile(lds, bh, bl, (size_t)z * Ncp + jt * KT, tid, wave);
        if (CAND && tid < KT)
            reinterpret_cast<float*>(lds + 2 * TILE_PLANE)[tid] = othr[(size_t)z * Ncp + jt * KT + tid];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x16 acc[2];
        score_tile(acc, lds, rf, l31, h);
        const bool special = (jt + 1) * KT > Nc || cmask != nullptr || rmask != nullptr;      // wave-uniform
        float tm = -HUGE_F;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[ct][r] * c1;
                if (special) {
                    const int j = jt * KT + 32 * ct + mfma32_row(r, h);
                    if (j >= Nc) x = -HUGE_F;
                    else if (rmasked || (cmask && !cmask[(size_t)z * Nc + j])) x = fill2;
                }
                acc[ct][r] = x;
                tm = fmaxf(tm, x);
            }
        // The tile's 32 terms are summed on their own and then added to the running sum with Kahan compensation:
        // once the row maximum (a term equal to 1) is in the accumulator, the other terms (~1e-8 each for a
        // confident match) are below half an ulp of it and a plain fp32 running sum would drop them one by one
        // (the "swamping" that costs the fp32 reference ~7e-5 on conf, oracle/coarse.py).
        const float mn = fmaxf(m, tm);
        const float resc = __builtin_amdgcn_exp2f(m - mn);
        float t = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += __builtin_amdgcn_exp2f(acc[ct][r] - mn);
        sum *= resc;
        comp *= resc;
        const float y = t - comp;
        const float ns = sum + y;
        comp = (ns - sum) - y;
        sum = ns;
        m = mn;
        if (CAND) {
            // log2 conf(i, j) = (x - rowmax_i - log2 rowsum_i) + (x - colmax_j - log2 colsum_j); this lane's running
            // (m, sum) -- this tile included -- bound the column term from ABOVE (final 2^M Sigma >= running 2^m sum), so
            //   2 x - (m + log2 sum) >= othr[i] = rowmax_i + log2 rowsum_i - 12
            // holds for every entry with conf >= 2^-12 (false positives only while the running sum is still small).
            const float clog = m + __builtin_amdgcn_logf(sum - comp);
            float hot = -1.f;
            const float* tl = reinterpret_cast<const float*>(lds + 2 * TILE_PLANE);      // this tile's 64 thresholds
            // streamed (no 32-register threshold array: the kernel sits at the 256-VGPR limit with its row fragments)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 a = *reinterpret_cast<const float4*>(tl + 32 * ct + 8 * q4 + 4 * h);
                    hot = fmaxf(hot, fmaxf(fmaxf(2.0f * acc[ct][4 * q4 + 0] - a.x, 2.0f * acc[ct][4 * q4 + 1] - a.y),
                                           fmaxf(2.0f * acc[ct][4 * q4 + 2] - a.z, 2.0f * acc[ct][4 * q4 + 3] - a.w)));
                }
            if (__builtin_amdgcn_ballot_w64(hot >= clog) != 0ull) {           // a candidate somewhere in this wave's tile
                uint2* const slots = cand + ((size_t)z * Nr + min(irow, Nr - 1)) * (2 * CAND_SLOTS) + h * CAND_SLOTS;
#pragma unroll 1
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = jt * KT + 32 * ct + mfma32_row(r, h);
                        const float xv = ct == 0 ? acc[0][r] : acc[1][r];
                        if (2.0f * xv >= tl[32 * ct + mfma32_row(r, h)] + clog && xv > -1.0e8f && i < Nc && irow < Nr) {
                            if (ncand < CAND_SLOTS) slots[ncand] = make_uint2((unsigned)i, __float_as_uint(xv));
                            ++ncand;
                        }
                    }
            }
        }
    }
    if (CAND && irow < Nr) cand_count[((size_t)z * Nr + irow) * 2 + h] = ncand;
    sum -= comp;
    const float mo = shfl_xor_f(m, 32), so = shfl_xor_f(sum, 32);
    const float mn = fmaxf(m, mo);
    const float st = sum * __builtin_amdgcn_exp2f(m - mn) + so * __builtin_amdgcn_exp2f(mo - mn);
    if (h == 0) {
        if (irow < Nr) stat[(size_t)z * Nr + irow] = make_float2(mn, st);
        if (dmax) {
            dmax[(size_t)z * Nrp + irow] = irow < Nr ? mn : HUGE_F;
            dinv[(size_t)z * Nrp + irow] = irow < Nr ? 1.0f / st : 0.f;
        }
        if (dthr) dthr[(size_t)z * Nrp + irow] = irow < Nr ? mn + __builtin_amdgcn_logf(st) - 12.0f : HUGE_F;
    }
}

// Tile prescreen of the match pass (round 4).  An entry can be a match only with conf > thr, and conf <= 2^(2x - rowmax_i - colmax_j)
// (both softmax sums are >= 1).  The hi.hi product alone bounds x from above: x <= x_hh + e, e = c1 |a|max |b|max (2^-10 (1 + 2^-11) +
// 2^-13) -- the two cross terms are at most 2^-11 |a_k| |b_k| each (Cauchy-Schwarz over the channels), 2^-13 covers the 768 fp32
// accumulation steps (4.6e-5); |a|max, |b|max from k1_prep.  tmask[z][row block of 128][tile][wave] = 1 when the wave's 32 rows x 64
// columns hold an entry with 2 (x_hh + e) - rowmax - colmax > log2 thr; k1_match skips the others: their entries cannot pass `> thr`,
// so neither the row bests that matter nor the column maxima (entries above thr only) change -- the same matches bit for bit (the
// surviving tiles are computed by score_tile as before: the SAME x the statistics passes summed), at a third of the matrix
// instructions and half the tile traffic for the 97 % of the tiles without a candidate (bench workload, tools/prescreen_stats.py).
__global__ __launch_bounds__(256, 2) void k1_screen(const _Float16* __restrict__ ah, const _Float16* __restrict__ bh, int Z, int L, int S,
                                                    int Lp, int Sp, float c1, const float2* __restrict__ rowstat,
                                                    const float* __restrict__ cmax, float thr, const unsigned* __restrict__ nmax,
                                                    uint8_t* __restrict__ tmask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nI = Lp / 128;
    int z, Ib;
    tile_coords(nI, Z, z, Ib);
    const int irow = Ib * 128 + 32 * wave + l31;
    f16x8 rh[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        rh[s] = *reinterpret_cast<const f16x8*>(ah + ((size_t)z * Lp + irow) * C + 8 * ((2 * s + h) ^ (irow & 15)));
    const float rmax = irow < L ? rowstat[(size_t)z * L + irow].x : HUGE_F;
    const float c2 = 2.0f * c1;
    const float slack2 = c2 * __builtin_sqrtf(__uint_as_float(nmax[0]) * __uint_as_float(nmax[1])) *
                         (1.0f / 1024.0f * (1.0f + 1.0f / 2048.0f) + 1.0f / 8192.0f);
    const float lim = __builtin_amdgcn_logf(thr) + rmax - slack2;       // candidate: 2 x_hh - colmax > lim
    const int ntile = (S + KT - 1) / KT;
    for (int jt = 0; jt < ntile; ++jt) {
        __syncthreads();
        {                                                               // the hi plane of the tile only
            const unsigned char* sh = reinterpret_cast<const unsigned char*>(bh + ((size_t)z * Sp + jt * KT) * C) + tid * 16;
#pragma unroll
            for (int j = 0; j < TILE_PLANE / 4096; ++j)
                __builtin_amdgcn_global_load_lds((gptr_t)(sh + j * 4096), (lptr_t)(lds + j * 4096 + wave * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int row = 32 * ct + l31;
                const f16x8 ch = *reinterpret_cast<const f16x8*>(lds + row * ROWB + (((2 * s + h) ^ (row & 15)) * 16));
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, rh[s], acc[ct], 0, 0, 0);
            }
        float top = -HUGE_F;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 a = *reinterpret_cast<const float4*>(cmax + (size_t)z * Sp + jt * KT + 32 * ct + 8 * q4 + 4 * h);
                top = fmaxf(top, fmaxf(fmaxf(fmaf(acc[ct][4 * q4 + 0], c2, -a.x), fmaf(acc[ct][4 * q4 + 1], c2, -a.y)),
                                       fmaxf(fmaf(acc[ct][4 * q4 + 2], c2, -a.z), fmaf(acc[ct][4 * q4 + 3], c2, -a.w))));
            }
        const bool can = !(top <= lim);                                 // also true for NaN
        const bool any = __builtin_amdgcn_ballot_w64(can) != 0ull;
        if (lane == 0) tmask[(((size_t)z * nI + Ib) * ntile + jt) * 4 + wave] = any ? 1 : 0;
    }
}

template <bool CONF>
__global__ __launch_bounds__(256, 2) void k1_match(const _Float16* __restrict__ ah, const _Float16* __restrict__ al,
                                                   const _Float16* __restrict__ bh, const _Float16* __restrict__ bl,
                                                   int Z, int L, int S, int Lp, int Sp, float c1, float fill2,
                                                   const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                   const float2* __restrict__ rowstat, const float* __restrict__ cmax,
                                                   const float* __restrict__ cinv, float* __restrict__ conf,
                                                   float* __restrict__ rowbest_v, int* __restrict__ rowbest_j,
                                                   unsigned* __restrict__ colbest, float thr, const uint8_t* __restrict__ tmask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nI = Lp / 128;
    int z, Ib;
    tile_coords(nI, Z, z, Ib);
    const int irow = Ib * 128 + 32 * wave + l31;
    const bool ivalid = irow < L;
    RowFrags rf;
    rf.load(ah, al, (size_t)z * Lp + irow, irow, h);
    const float2 rst = ivalid ? rowstat[(size_t)z * L + irow] : make_float2(0.f, 1.f);
    const float rinv = 1.0f / rst.y;
    const float c2 = 2.0f * c1;
    const bool rmasked = mask0 && ivalid && !mask0[(size_t)z * L + irow];
    const bool aligned4 = (S & 3) == 0;
    float bestv = -1.f;
    int bestj = 0x7fffffff;
    const int ntile = (S + KT - 1) / KT;
    for (int jt = 0; jt < ntile; ++jt) {
        // k1_screen's verdict on this workgroup's four 32-row blocks: no candidate in any of them -> nothing to do for the tile
        // (workgroup-uniform: the barriers below are skipped by all four waves together)
        if (tmask && *reinterpret_cast<const unsigned*>(tmask + (((size_t)z * nI + Ib) * ntile + jt) * 4) == 0u) continue;
        __syncthreads();                       // previous tile's fragments and column exchange consumed
        dma_tile(lds, bh, bl, (size_t)z * Sp + jt * KT, tid, wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tmask && tmask[(((size_t)z * nI + Ib) * ntile + jt) * 4 + wave] == 0) continue;      // this wave's 32 rows: no candidate
        f32x16 acc[2];
        score_tile(acc, lds, rf, l31, h);
        const bool masks = mask0 != nullptr || mask1 != nullptr;                 // wave-uniform
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            // column statistics of this lane's columns  j = jt*64 + 32 ct + 8 q + 4 h + (0..3)  (padded: +huge / 0 -> p = 0)
            float cm[16], ci[16];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const size_t o = (size_t)z * Sp + jt * KT + 32 * ct + 8 * q4 + 4 * h;
                const float4 a = *reinterpret_cast<const float4*>(cmax + o), b = *reinterpret_cast<const float4*>(cinv + o);
                cm[4 * q4 + 0] = a.x; cm[4 * q4 + 1] = a.y; cm[4 * q4 + 2] = a.z; cm[4 * q4 + 3] = a.w;
                ci[4 * q4 + 0] = b.x; ci[4 * q4 + 1] = b.y; ci[4 * q4 + 2] = b.z; ci[4 * q4 + 3] = b.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * KT + 32 * ct + mfma32_row(r, h);
                float x2 = acc[ct][r] * c2;                                       // 2 x in the log2 domain
                if (masks && j < S && (rmasked || (mask1 && !mask1[(size_t)z * S + j]))) x2 = 2.0f * fill2;
                const float p = __builtin_amdgcn_exp2f((x2 - rst.x) - cm[r]) * rinv * ci[r];
                const bool valid = ivalid && j < S;
                acc[ct][r] = valid ? p : -1.f;
                if (valid && p > bestv) { bestv = p; bestj = j; }
            }
            if (CONF && ivalid) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int j = jt * KT + 32 * ct + 8 * q4 + 4 * h;
                    float* dst = conf + ((size_t)z * L + irow) * S + j;
                    if (aligned4 && j + 3 < S) {
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[ct][4 * q4], acc[ct][4 * q4 + 1], acc[ct][4 * q4 + 2], acc[ct][4 * q4 + 3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (j + e < S) dst[e] = acc[ct][4 * q4 + e];
                    }
                }
            }
            // Column maxima, only where they can matter.  The mutual-nearest-neighbour test of k_finalize compares a row's
            // best confidence with its column's maximum, and only for rows whose best exceeds `thr`; a competitor in that
            // column beats it only with a confidence > thr as well.  So instead of the maximum of EVERY column over this
            // wave's 32 rows (32 cross-lane butterflies + an LDS exchange + a barrier per tile: more VALU work than the
            // tile's 96 MFMAs take) the entries above thr -- at most four per row and column, one per matched row in
            // practice -- go to colbest[z][j] by atomic max (confidences are positive floats: their bit patterns order
            // like unsigned integers; the array is zeroed by the caller); order-independent, hence deterministic.
            float tmax = fmaxf(fmaxf(acc[ct][0], acc[ct][1]), fmaxf(acc[ct][2], acc[ct][3]));
#pragma unroll
            for (int r = 4; r < 16; r += 2) tmax = fmaxf(tmax, fmaxf(acc[ct][r], acc[ct][r + 1]));
            if (__builtin_amdgcn_ballot_w64(tmax > thr) != 0ull) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc[ct][r] > thr)
                        atomicMax(colbest + (size_t)z * S + jt * KT + 32 * ct + mfma32_row(r, h), __float_as_uint(acc[ct][r]));
            }
        }
    }
    // merge the two half-waves (same rows, interleaved columns): larger value, ties -> smaller j
    const float vo = shfl_xor_f(bestv, 32);
    const int jo = shfl_xor_i(bestj, 32);
    if (vo > bestv || (vo == bestv && jo < bestj)) { bestv = vo; bestj = jo; }
    if (h == 0 && ivalid) {
        rowbest_v[(size_t)z * L + irow] = bestv;
        rowbest_j[(size_t)z * L + irow] = bestj;
    }
}

struct Ws16 {
    K1Workspace k;
    _Float16 *ah, *al, *bh, *bl;
    float2* colstat2;
    float *cmax, *cinv;
    float* rthr;             // [Z][Lp]: rowmax + log2(rowsum) - 12 (candidate threshold of the conf_matrix writer)
    uint8_t* tmask;          // [Z][Lp / 128][tiles of 64 columns][4 waves]: k1_screen's verdicts for k1_match
    unsigned* nmax;          // [2]: bit patterns of max row |2^4 f0|^2, max row |2^4 f1|^2 (prescreen bound of k1_match)
    int* fix_count;          // [Z][S][2]: entries each (column, half-wave) wanted to list for the writer's exact pass
    uint2* fix_list;         // [Z][S][2][CAND_SLOTS]: (i, bits of x)
    size_t bytes;
};
inline Ws16 carve16(void* ws, int Z, int L, int S) {
    Ws16 w;
    w.k = carve(ws, Z, L, S);
    const int Lp = (L + 127) / 128 * 128, Sp = (S + 127) / 128 * 128;
    unsigned char* p = (unsigned char*)ws;
    size_t o = w.k.bytes;
    auto take = [&](size_t n) { unsigned char* r = p ? p + o : nullptr; o += align256(n); return r; };
    w.ah = (_Float16*)take((size_t)Z * Lp * C * 2); w.al = (_Float16*)take((size_t)Z * Lp * C * 2);
    w.bh = (_Float16*)take((size_t)Z * Sp * C * 2); w.bl = (_Float16*)take((size_t)Z * Sp * C * 2);
    w.colstat2 = (float2*)take((size_t)Z * S * 8);
    w.cmax = (float*)take((size_t)Z * Sp * 4); w.cinv = (float*)take((size_t)Z * Sp * 4);
    w.rthr = (float*)take((size_t)Z * Lp * 4);
    w.nmax = (unsigned*)take(2 * sizeof(unsigned));
    w.tmask = (uint8_t*)take((size_t)Z * (Lp / 128) * ((S + KT - 1) / KT) * 4);
    w.fix_count = (int*)take((size_t)Z * S * 2 * sizeof(int));
    w.fix_list = (uint2*)take((size_t)Z * S * 2 * CAND_SLOTS * sizeof(uint2));
    w.bytes = o;
    return w;
}

}  // namespace

// ---- glue for the training kernels (dual_softmax_bwd_f16.hip): the forward workspace and its statistics passes ----
size_t far_k1_fwd_ws_bytes(int Z, int L, int S) { return carve16(nullptr, Z, L, S).bytes; }

static float k1_c1(float temperature) {
    return (float)(1.4426950408889634 / ((double)C * (double)temperature * PRESCALE * PRESCALE));
}

// operand planes + row statistics + dense column statistics, exactly as far_coarse_match_f16s prepares them
int far_k1_stats_launch(const float* f0, const float* f1, int Z, int L, int S, float temperature, void* ws, int* overflow,
                        hipStream_t stream) {
    const Ws16 w = carve16(ws, Z, L, S);
    const int Lp = (L + 127) / 128 * 128, Sp = (S + 127) / 128 * 128;
    const float c1 = k1_c1(temperature), fill2 = -1e9f * 1.44269504088896341f;
    auto gridp = [](long n) { long b = (n + 255) / 256; return (unsigned)(b < 65536L * 4 ? b : 65536L * 4); };
    hipLaunchKernelGGL(k1_prep, dim3(gridp((long)Z * Lp * 32)), dim3(256), 0, stream, f0, Z, L, Lp, w.ah, w.al, overflow, (unsigned*)nullptr);
    hipLaunchKernelGGL(k1_prep, dim3(gridp((long)Z * Sp * 32)), dim3(256), 0, stream, f1, Z, S, Sp, w.bh, w.bl, overflow, (unsigned*)nullptr);
    const size_t smem_s = 2 * TILE_PLANE + KT * sizeof(float);
    FAR_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)k1_rowstats<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_s));
    hipLaunchKernelGGL(k1_rowstats<false>, dim3((Lp / 128) * Z), dim3(256), smem_s, stream, w.ah, w.al, w.bh, w.bl, Z, L, S, Lp, Sp,
                       c1, fill2, (const uint8_t*)nullptr, (const uint8_t*)nullptr, w.k.rowstat, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (const float*)nullptr, (int*)nullptr, (uint2*)nullptr);
    hipLaunchKernelGGL(k1_rowstats<false>, dim3((Sp / 128) * Z), dim3(256), smem_s, stream, w.bh, w.bl, w.ah, w.al, Z, S, L, Sp, Lp,
                       c1, fill2, (const uint8_t*)nullptr, (const uint8_t*)nullptr, w.colstat2, w.cmax, w.cinv, (float*)nullptr,
                       (const float*)nullptr, (int*)nullptr, (uint2*)nullptr);
    return far_check_launch();
}

void far_k1_fwd_views(void* ws, int Z, int L, int S, const _Float16** ah, const _Float16** bh, const float2** rowstat,
                      const float** cmax, const float** cinv, float* c1_out) {
    const Ws16 w = carve16(ws, Z, L, S);
    *ah = w.ah; *bh = w.bh; *rowstat = w.k.rowstat; *cmax = w.cmax; *cinv = w.cinv;
    *c1_out = 0.f;      // set by the caller from the temperature (kept out of the workspace: it is not device data)
}

extern "C" {

size_t far_coarse_match_f16s_workspace_bytes(int Z, int L, int S, int Cc) {
    if (Z <= 0 || L <= 0 || S <= 0 || Cc != C) return 0;
    return carve16(nullptr, Z, L, S).bytes;
}

// Split-fp16 variant of far_coarse_match_f32: same arguments, same outputs (C must be 256).
int far_coarse_match_f16s(const float* f0, const float* f1, int Z, int L, int S, int Cc,
                          float temperature, float thr, int border, int h0, int w0, int h1, int w1,
                          float cell_scale, const uint8_t* mask0, const uint8_t* mask1,
                          const int* valid_hw, const float* scale0, const float* scale1,
                          float* conf_out, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                          float* mkpts0_c, float* mkpts1_c, int* counts_out, int* total_out,
                          void* ws, int* overflow, hipStream_t stream) {
    far_clear_errors();
    if (!f0 || !f1 || !ws || !b_ids || !i_ids || !j_ids || !mconf || !mkpts0_c || !mkpts1_c || !total_out)
        return FAR_EINVAL;
    if (Z <= 0 || L <= 0 || S <= 0 || Cc != C || h0 * w0 != L || h1 * w1 != S || (long)Z * (L > S ? L : S) > 0x7ff00000L) return FAR_EINVAL;
    const Ws16 w = carve16(ws, Z, L, S);
    const int Lp = (L + 127) / 128 * 128, Sp = (S + 127) / 128 * 128;
    // sim = <f0 / sqrt(C), f1 / sqrt(C)> / temperature  ->  log2 domain, operands pre-scaled by 2^4 each
    const float c1 = (float)(1.4426950408889634 / ((double)C * (double)temperature * PRESCALE * PRESCALE));
    const float fill2 = -1e9f * 1.44269504088896341f;
    auto gridp = [](long n) { long b = (n + 255) / 256; return (unsigned)(b < 65536L * 4 ? b : 65536L * 4); };
    const bool screen = !conf_out && !mask0 && !mask1 && far_get_tuning(10) == 0;      // the match pass's tile prescreen (tuning 10: 1 = off)
    if (screen) hipMemsetAsync(w.nmax, 0, 2 * sizeof(unsigned), stream);
    hipLaunchKernelGGL(k1_prep, dim3(gridp((long)Z * Lp * 32)), dim3(256), 0, stream, f0, Z, L, Lp, w.ah, w.al, overflow, screen ? w.nmax : (unsigned*)nullptr);
    hipLaunchKernelGGL(k1_prep, dim3(gridp((long)Z * Sp * 32)), dim3(256), 0, stream, f1, Z, S, Sp, w.bh, w.bl, overflow, screen ? w.nmax + 1 : (unsigned*)nullptr);
    int* counts = counts_out ? counts_out : w.k.counts;
    hipMemsetAsync(counts, 0, sizeof(int) * Z, stream);
    const size_t smem_s = 2 * TILE_PLANE + KT * sizeof(float), smem_m = 2 * TILE_PLANE + 4 * 64 * sizeof(float);
    FAR_ONCE_PER_DEVICE(
        hipFuncSetAttribute((const void*)k1_rowstats<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_s);
        hipFuncSetAttribute((const void*)k1_match<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_m);
        hipFuncSetAttribute((const void*)k1_match<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_m));
    hipLaunchKernelGGL(k1_rowstats<false>, dim3((Lp / 128) * Z), dim3(256), smem_s, stream, w.ah, w.al, w.bh, w.bl, Z, L, S, Lp, Sp,
                       c1, fill2, mask0, mask1, w.k.rowstat, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       (const float*)nullptr, (int*)nullptr, (uint2*)nullptr);
    hipLaunchKernelGGL(k1_rowstats<false>, dim3((Sp / 128) * Z), dim3(256), smem_s, stream, w.bh, w.bl, w.ah, w.al, Z, S, L, Sp, Lp,
                       c1, fill2, mask1, mask0, w.colstat2, w.cmax, w.cinv, (float*)nullptr, (const float*)nullptr,
                       (int*)nullptr, (uint2*)nullptr);
    const int nI = Lp / 128;
    hipMemsetAsync(w.k.colbest_part, 0, sizeof(float) * (size_t)Z * S, stream);      // the atomic column maxima start at 0
    if (screen) {
        FAR_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)k1_screen, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_PLANE));
        hipLaunchKernelGGL(k1_screen, dim3(nI * Z), dim3(256), TILE_PLANE, stream, w.ah, w.bh, Z, L, S, Lp, Sp, c1, w.k.rowstat, w.cmax, thr,
                           (const unsigned*)w.nmax, w.tmask);
    }
    if (conf_out)
        hipLaunchKernelGGL(k1_match<true>, dim3(nI * Z), dim3(256), smem_m, stream, w.ah, w.al, w.bh, w.bl, Z, L, S, Lp, Sp, c1,
                           fill2, mask0, mask1, w.k.rowstat, w.cmax, w.cinv, conf_out, w.k.rowbest_v, w.k.rowbest_j, reinterpret_cast<unsigned*>(w.k.colbest_part), thr,
                           screen ? (const uint8_t*)w.tmask : (const uint8_t*)nullptr);
    else
        hipLaunchKernelGGL(k1_match<false>, dim3(nI * Z), dim3(256), smem_m, stream, w.ah, w.al, w.bh, w.bl, Z, L, S, Lp, Sp, c1,
                           fill2, mask0, mask1, w.k.rowstat, w.cmax, w.cinv, conf_out, w.k.rowbest_v, w.k.rowbest_j, reinterpret_cast<unsigned*>(w.k.colbest_part), thr,
                           screen ? (const uint8_t*)w.tmask : (const uint8_t*)nullptr);
    // colbest_part doubles as the single [Z][S] array of atomic column maxima (nI = 1 for k_finalize)
    hipLaunchKernelGGL(k_finalize, dim3((L + 255) / 256, Z), dim3(256), 0, stream, w.k.rowbest_v, w.k.rowbest_j,
                       w.k.colbest_part, 1, L, S, thr, border, h0, w0, h1, w1, valid_hw, w.k.match_j, counts);
    hipLaunchKernelGGL(k_compact, dim3(Z), dim3(256), 0, stream, w.k.match_j, w.k.rowbest_v, counts, L, w0, w1,
                       cell_scale, scale0, scale1, b_ids, i_ids, j_ids, mconf, mkpts0_c, mkpts1_c, total_out);
    return far_check_launch();
}

// data['conf_matrix'] alone (coarse_matching.py:108-118), at HBM write speed: statistics on the split-precision passes
// (fp32-grade), the matrix itself from plain-fp16 scores, and every entry whose row softmax exceeds 2^-12 (listed by the
// second statistics pass together with its split-precision score) rewritten with the fused matcher's exact formula
// (dual_softmax_conf_f16.hip).  stages: bit 0 = operand planes + statistics, bit 1 = write the matrix (a caller that
// keeps the workspace may run the two separately, e.g. to time the writer alone).  fix_info_out: optional 2 device
// ints = (entries listed for the exact recomputation, list capacity): listed > capacity means the surplus entries kept
// their fp16-operand value (relative error <= ~1e-3); the Python front end then falls back to far_coarse_match_f16s.
int far_conf_matrix_f16s(const float* f0, const float* f1, int Z, int L, int S, int Cc, float temperature,
                         const uint8_t* mask0, const uint8_t* mask1, int stages, float* conf_out, int* fix_info_out,
                         void* ws, int* overflow, hipStream_t stream) {
    far_clear_errors();
    if (!f0 || !f1 || !ws || !conf_out || Z <= 0 || L <= 0 || S <= 0 || Cc != C || !(stages & 3) || (long)Z * (L > S ? L : S) > 0x7ff00000L) return FAR_EINVAL;
    const Ws16 w = carve16(ws, Z, L, S);
    const int Lp = (L + 127) / 128 * 128, Sp = (S + 127) / 128 * 128;
    const float c1 = (float)(1.4426950408889634 / ((double)C * (double)temperature * PRESCALE * PRESCALE));
    const float fill2 = -1e9f * 1.44269504088896341f;
    if (stages & 1) {
        auto gridp = [](long n) { long b = (n + 255) / 256; return (unsigned)(b < 65536L * 4 ? b : 65536L * 4); };
        hipLaunchKernelGGL(k1_prep, dim3(gridp((long)Z * Lp * 32)), dim3(256), 0, stream, f0, Z, L, Lp, w.ah, w.al, overflow, (unsigned*)nullptr);
        hipLaunchKernelGGL(k1_prep, dim3(gridp((long)Z * Sp * 32)), dim3(256), 0, stream, f1, Z, S, Sp, w.bh, w.bl, overflow, (unsigned*)nullptr);
        const size_t smem_s = 2 * TILE_PLANE + KT * sizeof(float);
        FAR_ONCE_PER_DEVICE(
            hipFuncSetAttribute((const void*)k1_rowstats<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_s);
            hipFuncSetAttribute((const void*)k1_rowstats<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_s));
        // row statistics (+ the dense candidate thresholds), then column statistics with candidate listing
        hipLaunchKernelGGL(k1_rowstats<false>, dim3((Lp / 128) * Z), dim3(256), smem_s, stream, w.ah, w.al, w.bh, w.bl, Z, L, S, Lp,
                           Sp, c1, fill2, mask0, mask1, w.k.rowstat, (float*)nullptr, (float*)nullptr, w.rthr,
                           (const float*)nullptr, (int*)nullptr, (uint2*)nullptr);
        hipLaunchKernelGGL(k1_rowstats<true>, dim3((Sp / 128) * Z), dim3(256), smem_s, stream, w.bh, w.bl, w.ah, w.al, Z, S, L, Sp,
                           Lp, c1, fill2, mask1, mask0, w.colstat2, w.cmax, w.cinv, (float*)nullptr, (const float*)w.rthr,
                           w.fix_count, w.fix_list);
    }
    if (stages & 2) {
        const int rc = far_k1_conf_launch(w.ah, w.bh, Z, L, S, Lp, Sp, c1, fill2, mask0, mask1, w.k.rowstat, w.cmax, w.cinv,
                                          conf_out, w.fix_count, w.fix_list, CAND_SLOTS, fix_info_out, stream);
        if (rc != FAR_OK) return rc;
    }
    return far_check_launch();
}

}  // extern "C"
