// K1 (coarse matcher) and the statistics half of K2 (EMM head): all-pairs correlation with
// dual-softmax, never materialising the L x S score matrix unless asked to.
//
// Replaces (reference, mp3d_loftr/src/loftr/utils/coarse_matching.py):
//   :104-113  feat / sqrt(C); sim = einsum(nlc,nsc->nls) / temperature; optional -INF mask fill
//   :118      conf = softmax(sim, 1) * softmax(sim, 2)
//   :174-195  conf > thr, border removal, mutual nearest neighbour, first-True j per row, mconf
//   :246-263  coordinates of the matched cells
// and (mp3d_loftr/src/loftr/loftr_module/transformer.py:278-282) the two softmaxes of the head.
//
// Pass structure (fp32-exact variant; every dot product is an fmaf chain on the f32 matrix core):
//   stats    : tile S = f0 f1^T, online row (max,sum-exp) per lane, per-tile column (max,sum-exp)
//              partials -> colpart[n][Iblk][j]                                    (1 GEMM)
//   colreduce: colstat[n][j] = merge over Iblk
//   match    : recompute the tile, P = softmax_col * softmax_row, optional conf_matrix store,
//              running row best (P, j) and per-tile column best P -> colbest[n][Iblk][j]  (1 GEMM)
//   finalize : per row: P>thr, border(i), border(j*), P == column max of P -> match_j[n][i]
//   compact  : ordered (b, i) compaction to int64 ids, conf and cell coordinates
#include "gemm_tile_f32.h"

namespace {

constexpr float NEG_BIG = -FLT_MAX;

struct SimParams {
    float feat_div;   // features divided by this when staged; 1 when the scaling is folded into acc_scale
    float acc_scale;  // exact power-of-two factor applied to the dot product (1/feat_div^2 when folded, else 1)
    float sim_div;    // then / sim_div           (temperature for K1, 1 for K2)
    float sim_rcp;    // 1 / sim_div (IEEE), for the 3-instruction exact division
    float sim_mul;    // then * sim_mul           (1 for K1, head_dim^-0.5 for K2)
    float mask_fill;  // value for masked-out (i,j) pairs (-1e9 in the reference)
    int stagger;      // tuning: wave-slot priority staggering on/off
    float k2;         // acc_scale / sim_div * sim_mul * log2(e): log2-domain score per unit dot product (bf16 path)
};

// The reference divides both feature maps by sqrt(C) before the contraction (coarse_matching.py:104-105).  When
// sqrt(C) is a power of two that scaling commutes exactly with every rounding of the fmaf chain, so it is applied
// once to the accumulator instead of to 2 x 32 floats per thread per chunk: bit-identical, far fewer instructions.
inline SimParams make_sim(float feat_div, float sim_div, float sim_mul) {
    SimParams p;
    int e;
    float m = frexpf(feat_div, &e);
    bool pow2 = (m == 0.5f);
    p.feat_div = pow2 ? 1.0f : feat_div;
    p.acc_scale = pow2 ? 1.0f / (feat_div * feat_div) : 1.0f;
    p.sim_div = sim_div;
    p.sim_rcp = 1.0f / sim_div;
    p.sim_mul = sim_mul;
    p.mask_fill = -1e9f;
    p.stagger = 0;
    p.k2 = (float)((double)p.acc_scale / (double)sim_div * (double)sim_mul * 1.4426950408889634);
    return p;
}

__device__ __forceinline__ float sim_of(float acc, const SimParams& p) {
    float s = fdiv_by(acc * p.acc_scale, p.sim_div, p.sim_rcp);
    return s * p.sim_mul;
}

// --------------------------------------------------------------------------------------------
// Pass 1: statistics.
// grid = (nI, Z); block = 256.  rowstat[z][i] = (max_j s, sum_j exp(s - max));
// colpart[z][Iblk][j] = (max over the block's rows, sum exp).
// mask0/mask1 (optional, uint8 [Z][L] / [Z][S]): pair (i,j) is masked iff !(mask0[i] && mask1[j]).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_stats_f32(
    const float* __restrict__ f0, const float* __restrict__ f1, int Z, int L, int S, int C, SimParams sp,
    const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
    float2* __restrict__ rowstat, float2* __restrict__ colpart) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    TileLds& lds = *reinterpret_cast<TileLds*>(smem_raw);
    float2* colx = reinterpret_cast<float2*>(smem_raw + sizeof(TileLds));  // [4][128]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nI = (L + TILE_M - 1) / TILE_M;
    int z, Ib;
    tile_coords(nI, Z, z, Ib);
    const int i0 = Ib * TILE_M;
    const float* A = f0 + (size_t)z * L * C;
    const float* B = f1 + (size_t)z * S * C;
    const int nJ = (S + TILE_N - 1) / TILE_N, nkc = C / KC, nq = nJ * nkc;

    // row masks for this lane's 16 rows
    unsigned rowvalid = 0, rowmasked = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int i = i0 + 32 * wave + mfma32_row(r, h);
        if (i < L) {
            rowvalid |= 1u << r;
            if (mask0 && !mask0[(size_t)z * L + i]) rowmasked |= 1u << r;
        }
    }

    stagger_priority_by_wave_slot(sp.stagger);
    float rm[16], rs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { rm[r] = NEG_BIG; rs[r] = 0.f; }

    f32x16 acc[4];
    acc_zero(acc);
    ChunkRegs cr;
    chunk_load(cr, A, i0, L, B, 0, S, C, 0, tid);
    chunk_store(cr, lds, 0, tid, sp.feat_div);
    __syncthreads();

    for (int q = 0; q < nq; ++q) {
        const int Jt = q / nkc, kc = q - Jt * nkc;
        const int buf = q & 1;
        if (q + 1 < nq) {
            int Jn = (q + 1) / nkc, kn = (q + 1) - Jn * nkc;
            chunk_load(cr, A, i0, L, B, Jn * TILE_N, S, C, kn * KC, tid);
        }
        chunk_mfma<false>(acc, lds, buf, wave, lane);
        if (kc == nkc - 1) {
            // ---- tile epilogue ----
            const int j0 = Jt * TILE_N;
            float cmx[4], csm[4];
            bool cvalid[4], cmasked[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                int j = j0 + 32 * ct + l31;
                cvalid[ct] = j < S;
                cmasked[ct] = cvalid[ct] && mask1 && !mask1[(size_t)z * S + j];
            }
            // similarity values in place
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = sim_of(acc[ct][r], sp);
                    if (cmasked[ct] || ((rowmasked >> r) & 1)) s = sp.mask_fill;
                    acc[ct][r] = s;
                }
            // row direction: online update over this lane's 4 columns
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float tm = NEG_BIG;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    if (cvalid[ct]) tm = fmaxf(tm, acc[ct][r]);
                float mn = fmaxf(rm[r], tm);
                float sum = (rm[r] == mn) ? rs[r] : rs[r] * fexp(rm[r] - mn);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    if (cvalid[ct]) sum += fexp(acc[ct][r] - mn);
                rm[r] = mn;
                rs[r] = sum;
            }
            // column direction: reduce this lane's 16 rows, then the two lane halves
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                float m = NEG_BIG;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((rowvalid >> r) & 1) m = fmaxf(m, acc[ct][r]);
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((rowvalid >> r) & 1) s += fexp(acc[ct][r] - m);
                float mo = shfl_xor_f(m, 32), so = shfl_xor_f(s, 32);
                softmax_merge(m, s, mo, so);
                cmx[ct] = m;
                csm[ct] = s;
            }
            if (h == 0) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) colx[wave * 128 + 32 * ct + l31] = make_float2(cmx[ct], csm[ct]);
            }
            acc_zero(acc);
        }
        if (q + 1 < nq) chunk_store(cr, lds, buf ^ 1, tid, sp.feat_div);
        __syncthreads();
        if (kc == nkc - 1) {
            const int j0 = Jt * TILE_N;
            if (tid < 128 && j0 + tid < S) {
                float2 v = colx[tid];
                float m = v.x, s = v.y;
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    float2 o = colx[w * 128 + tid];
                    softmax_merge(m, s, o.x, o.y);
                }
                colpart[((size_t)z * nI + Ib) * S + j0 + tid] = make_float2(m, s);
            }
            // colx is rewritten only after the next tile's nkc >= 1 barriers: no extra sync needed
            // when nkc >= 2; for nkc == 1 add one.
            if (nkc == 1) __syncthreads();
        }
    }

    // merge row partials over the 32 lanes that share (wave, h)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float m = rm[r], s = rs[r];
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            float mo = shfl_xor_f(m, d), so = shfl_xor_f(s, d);
            softmax_merge(m, s, mo, so);
        }
        if (l31 == 0 && ((rowvalid >> r) & 1)) {
            int i = i0 + 32 * wave + mfma32_row(r, h);
            rowstat[(size_t)z * L + i] = make_float2(m, s);
        }
    }
}

// --------------------------------------------------------------------------------------------
// Pass 1, 64-channel variant (the head's q/k, C = 64, no masks, no feature scaling).
// The row-side panel (32 rows x 64 channels per wave) lives in registers for the whole sweep; only the column-side
// tile is staged in LDS (128 x 64 floats, single buffer, 35 KB), processed as two 64-column halves with two
// accumulators.  ~120 VGPRs and 39 KB LDS -> 4 workgroups per CU, so the statistics VALU work of one wave runs in
// the MFMA shadow of three others.  Same outputs as k_stats_f32.
// --------------------------------------------------------------------------------------------
constexpr int S64_KS = 68;   // LDS row stride (floats): 17 16-byte slots, odd -> conflict-free ds_read_b128

__global__ __launch_bounds__(256, 3) void k_stats_c64_f32(
    const float* __restrict__ f0, const float* __restrict__ f1, int Z, int L, int S, SimParams sp,
    float2* __restrict__ rowstat, float2* __restrict__ colpart) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* kt = reinterpret_cast<float*>(smem_raw);                                   // [128][68]
    float2* colx = reinterpret_cast<float2*>(smem_raw + 128 * S64_KS * sizeof(float));  // [4][64]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nI = (L + TILE_M - 1) / TILE_M;
    int z, Ib;
    tile_coords(nI, Z, z, Ib);
    const int i0 = Ib * TILE_M;
    const float* A = f0 + (size_t)z * L * 64;
    const float* B = f1 + (size_t)z * S * 64;

    // A fragment: row i = i0 + 32 wave + l31, channels {8g + 4h .. +3}
    const int irow = i0 + 32 * wave + l31;
    f32x4 qa[8];
#pragma unroll
    for (int g = 0; g < 8; ++g)
        qa[g] = irow < L ? *reinterpret_cast<const f32x4*>(A + (size_t)irow * 64 + 8 * g + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned rowvalid = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (i0 + 32 * wave + mfma32_row(r, h) < L) rowvalid |= 1u << r;
    float rm[16], rs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { rm[r] = NEG_BIG; rs[r] = 0.f; }

    const int nJ = (S + TILE_N - 1) / TILE_N;
    for (int Jt = 0; Jt < nJ; ++Jt) {
        const int j0 = Jt * TILE_N;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int idx = tid + 256 * p, row = idx >> 4, slot = idx & 15;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j0 + row < S) kv = *reinterpret_cast<const float4*>(B + (size_t)(j0 + row) * 64 + slot * 4);
            *reinterpret_cast<float4*>(&kt[row * S64_KS + slot * 4]) = kv;
        }
        __syncthreads();
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int jb = 64 * half;
            f32x16 acc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
            const float* krow = &kt[(jb + l31) * S64_KS + 4 * h];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f32x4 kb[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) kb[ct] = *reinterpret_cast<const f32x4*>(krow + ct * 32 * S64_KS + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[g][c], kb[ct][c], acc[ct], 0, 0, 0);
            }
            // lane holds column j = j0 + jb + 32 ct + l31, rows i0 + 32 wave + mfma32_row(r, h)
            bool cvalid[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) cvalid[ct] = (j0 + jb + 32 * ct + l31) < S;
            if (sp.sim_div == 1.0f && sp.acc_scale == 1.0f) {      // the head: s = dot * scale, nothing else
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ct][r] = acc[ct][r] * sp.sim_mul;
            } else {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ct][r] = sim_of(acc[ct][r], sp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float tm = NEG_BIG;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    if (cvalid[ct]) tm = fmaxf(tm, acc[ct][r]);
                const float mn = fmaxf(rm[r], tm);
                float sum = (rm[r] == mn) ? rs[r] : rs[r] * fexp(rm[r] - mn);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    if (cvalid[ct]) sum += fexp(acc[ct][r] - mn);
                rm[r] = mn;
                rs[r] = sum;
            }
            float cmx[2], csm[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float m = NEG_BIG;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((rowvalid >> r) & 1) m = fmaxf(m, acc[ct][r]);
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((rowvalid >> r) & 1) s += fexp(acc[ct][r] - m);
                const float mo = shfl_xor_f(m, 32), so = shfl_xor_f(s, 32);
                softmax_merge(m, s, mo, so);
                cmx[ct] = m;
                csm[ct] = s;
            }
            __syncthreads();                        // colx free (previous half's readers are done)
            if (h == 0) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) colx[wave * 64 + 32 * ct + l31] = make_float2(cmx[ct], csm[ct]);
            }
            __syncthreads();
            if (tid < 64 && j0 + jb + tid < S) {
                float2 v = colx[tid];
                float m = v.x, s = v.y;
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float2 o = colx[w * 64 + tid];
                    softmax_merge(m, s, o.x, o.y);
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
            softmax_merge(m, s, mo, so);
        }
        if (l31 == 0 && ((rowvalid >> r) & 1))
            rowstat[(size_t)z * L + i0 + 32 * wave + mfma32_row(r, h)] = make_float2(m, s);
    }
}

// colstat[z][j] = merge_{Ib} colpart[z][Ib][j]
__global__ void k_colreduce(const float2* __restrict__ colpart, int nI, int S, float2* __restrict__ colstat) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, z = blockIdx.y;
    if (j >= S) return;
    float m = NEG_BIG, s = 0.f;
    for (int b = 0; b < nI; ++b) {
        float2 v = colpart[((size_t)z * nI + b) * S + j];
        softmax_merge(m, s, v.x, v.y);
    }
    colstat[(size_t)z * S + j] = make_float2(m, s);
}

// --------------------------------------------------------------------------------------------
// Pass 2: recompute, P = softmax_col * softmax_row, conf store (optional), row/col best.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_match_f32(
    const float* __restrict__ f0, const float* __restrict__ f1, int Z, int L, int S, int C, SimParams sp,
    const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
    const float2* __restrict__ rowstat, const float2* __restrict__ colstat,
    float* __restrict__ conf,  // optional [Z][L][S]
    float* __restrict__ rowbest_v, int* __restrict__ rowbest_j,  // [Z][L]
    float* __restrict__ colbest_part) {                          // [Z][nI][S]
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    TileLds& lds = *reinterpret_cast<TileLds*>(smem_raw);
    float* colx = reinterpret_cast<float*>(smem_raw + sizeof(TileLds));  // [4][128]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nI = (L + TILE_M - 1) / TILE_M;
    int z, Ib;
    tile_coords(nI, Z, z, Ib);
    const int i0 = Ib * TILE_M;
    const float* A = f0 + (size_t)z * L * C;
    const float* B = f1 + (size_t)z * S * C;
    const int nJ = (S + TILE_N - 1) / TILE_N, nkc = C / KC, nq = nJ * nkc;

    unsigned rowvalid = 0, rowmasked = 0;
    float rmax[16], rsum[16], bestv[16];
    int bestj[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int i = i0 + 32 * wave + mfma32_row(r, h);
        rmax[r] = 0.f; rsum[r] = 1.f; bestv[r] = -1.f; bestj[r] = 0x7fffffff;
        if (i < L) {
            rowvalid |= 1u << r;
            if (mask0 && !mask0[(size_t)z * L + i]) rowmasked |= 1u << r;
            float2 st = rowstat[(size_t)z * L + i];
            rmax[r] = st.x; rsum[r] = 1.0f / st.y;   // rsum holds the reciprocal of the row sum
        }
    }

    stagger_priority_by_wave_slot(sp.stagger);
    f32x16 acc[4];
    acc_zero(acc);
    ChunkRegs cr;
    chunk_load(cr, A, i0, L, B, 0, S, C, 0, tid);
    chunk_store(cr, lds, 0, tid, sp.feat_div);
    __syncthreads();

    for (int q = 0; q < nq; ++q) {
        const int Jt = q / nkc, kc = q - Jt * nkc;
        const int buf = q & 1;
        if (q + 1 < nq) {
            int Jn = (q + 1) / nkc, kn = (q + 1) - Jn * nkc;
            chunk_load(cr, A, i0, L, B, Jn * TILE_N, S, C, kn * KC, tid);
        }
        chunk_mfma<false>(acc, lds, buf, wave, lane);
        if (kc == nkc - 1) {
            const int j0 = Jt * TILE_N;
            float cbest[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int j = j0 + 32 * ct + l31;
                const bool cvalid = j < S;
                const bool cmasked = cvalid && mask1 && !mask1[(size_t)z * S + j];
                float2 cst = cvalid ? colstat[(size_t)z * S + j] : make_float2(0.f, 1.f);
                const float cinv = 1.0f / cst.y;
                float cb = -1.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = sim_of(acc[ct][r], sp);
                    if (cmasked || ((rowmasked >> r) & 1)) s = sp.mask_fill;
                    // reference order: softmax over dim 1 (columns normalised over rows) times
                    // softmax over dim 2 (coarse_matching.py:118)
                    // softmax = exp(s - max) * (1 / sum): the reciprocal is IEEE, the product is within 1 ulp of
                    // the reference's division
                    float pc = fexp(s - cst.x) * cinv;
                    float pr = fexp(s - rmax[r]) * rsum[r];
                    float p = pc * pr;
                    const bool ok = cvalid && ((rowvalid >> r) & 1);
                    if (ok) {
                        if (conf) {
                            int i = i0 + 32 * wave + mfma32_row(r, h);
                            conf[((size_t)z * L + i) * S + j] = p;
                        }
                        if (p > bestv[r]) { bestv[r] = p; bestj[r] = j; }
                        cb = fmaxf(cb, p);
                    }
                }
                cb = fmaxf(cb, shfl_xor_f(cb, 32));
                cbest[ct] = cb;
            }
            if (h == 0) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) colx[wave * 128 + 32 * ct + l31] = cbest[ct];
            }
            acc_zero(acc);
        }
        if (q + 1 < nq) chunk_store(cr, lds, buf ^ 1, tid, sp.feat_div);
        __syncthreads();
        if (kc == nkc - 1) {
            const int j0 = Jt * TILE_N;
            if (tid < 128 && j0 + tid < S) {
                float m = fmaxf(fmaxf(colx[tid], colx[128 + tid]), fmaxf(colx[256 + tid], colx[384 + tid]));
                colbest_part[((size_t)z * nI + Ib) * S + j0 + tid] = m;
            }
            if (nkc == 1) __syncthreads();
        }
    }

    // row best across the 32 lanes sharing (wave, h): larger P wins, ties -> smaller j
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = bestv[r];
        int j = bestj[r];
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            float vo = shfl_xor_f(v, d);
            int jo = shfl_xor_i(j, d);
            if (vo > v || (vo == v && jo < j)) { v = vo; j = jo; }
        }
        if (l31 == 0 && ((rowvalid >> r) & 1)) {
            int i = i0 + 32 * wave + mfma32_row(r, h);
            rowbest_v[(size_t)z * L + i] = v;
            rowbest_j[(size_t)z * L + i] = j;
        }
    }
}

// --------------------------------------------------------------------------------------------
// finalize: match_j[z][i] = j* if row i is a mutual-nearest match above threshold and inside the
// border, else -1.  Border semantics of mask_border (coarse_matching.py:8-25): a cell (y,x) of an
// h x w grid survives iff bd <= y < h-bd and bd <= x < w-bd.  With padded masks
// (mask_border_with_padding, :28-43) the lower limits come from per-sample valid extents hv/wv.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ bool border_ok(int idx, int w, int hlim, int wlim, int bd) {
    int y = idx / w, x = idx - y * w;
    return y >= bd && y < hlim - bd && x >= bd && x < wlim - bd;
}

__global__ void k_finalize(const float* __restrict__ rowbest_v, const int* __restrict__ rowbest_j,
                           const float* __restrict__ colbest_part, int nI, int L, int S, float thr,
                           int bd, int h0, int w0, int h1, int w1,
                           const int* __restrict__ valid_hw,  // optional [Z][4] = h0v,w0v,h1v,w1v
                           int* __restrict__ match_j, int* __restrict__ counts) {
    const int z = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int hl0 = h0, wl0 = w0, hl1 = h1, wl1 = w1;
    if (valid_hw) { hl0 = valid_hw[z * 4]; wl0 = valid_hw[z * 4 + 1]; hl1 = valid_hw[z * 4 + 2]; wl1 = valid_hw[z * 4 + 3]; }
    int mj = -1;
    if (i < L) {
        float v = rowbest_v[(size_t)z * L + i];
        int j = rowbest_j[(size_t)z * L + i];
        if (v > thr && j < S) {
            bool ok = bd <= 0 || (border_ok(i, w0, hl0, wl0, bd) && border_ok(j, w1, hl1, wl1, bd));
            if (ok) {
                float cm = -1.f;
                for (int b = 0; b < nI; ++b) cm = fmaxf(cm, colbest_part[((size_t)z * nI + b) * S + j]);
                if (v == cm) mj = j;
            }
        }
        match_j[(size_t)z * L + i] = mj;
    }
    unsigned long long bal = __ballot(mj >= 0);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&counts[z], __popcll(bal));
}

// compact: one block per pair; offset = sum of counts of earlier pairs; ordered by i.
__global__ void k_compact(const int* __restrict__ match_j, const float* __restrict__ rowbest_v,
                          const int* __restrict__ counts, int L, int w0, int w1, float scale,
                          const float* __restrict__ scale0, const float* __restrict__ scale1,  // optional [Z][2]
                          int64_t* __restrict__ b_ids, int64_t* __restrict__ i_ids,
                          int64_t* __restrict__ j_ids, float* __restrict__ mconf,
                          float* __restrict__ mkpts0, float* __restrict__ mkpts1,
                          int* __restrict__ total) {
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int z = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        int off = 0;
        for (int b = 0; b < z; ++b) off += counts[b];
        base_s = off;
        if (z == gridDim.x - 1) *total = off + counts[z];
    }
    __syncthreads();
    int base = base_s;
    for (int i0 = 0; i0 < L; i0 += 256) {
        int i = i0 + tid;
        int mj = (i < L) ? match_j[(size_t)z * L + i] : -1;
        unsigned long long bal = __ballot(mj >= 0);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_cnt[w];
        int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (mj >= 0) {
            int pos = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
            b_ids[pos] = z;
            i_ids[pos] = i;
            j_ids[pos] = mj;
            mconf[pos] = rowbest_v[(size_t)z * L + i];
            float sx0 = scale, sy0 = scale, sx1 = scale, sy1 = scale;
            if (scale0) { sx0 = scale * scale0[z * 2]; sy0 = scale * scale0[z * 2 + 1]; }
            if (scale1) { sx1 = scale * scale1[z * 2]; sy1 = scale * scale1[z * 2 + 1]; }
            mkpts0[2 * pos] = (float)(i % w0) * sx0;
            mkpts0[2 * pos + 1] = (float)(i / w0) * sy0;
            mkpts1[2 * pos] = (float)(mj % w1) * sx1;
            mkpts1[2 * pos + 1] = (float)(mj / w1) * sy1;
        }
        base += tot;
        __syncthreads();
    }
}

// ============================================================================================
// bf16-input variant of K1 (north_star: "MFMA bf16 where they are genuine dense contractions").
// Features are rounded once to bf16 (RNE); the contraction runs on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation; everything after the dot product (scaling, softmax statistics, P, selection) is the fp32 code of
// the exact variant.  Parity statement: bit-exact indices / 1e-5 confidences against the oracle evaluated on the
// SAME bf16-rounded features; against the fp32 path it is reported as match-set IoU (tests/test_coarse_gpu.py).
// The row panel (32 rows x C channels per wave) is register resident (C/4 VGPRs); the 128-column tile of the
// other map is staged once per tile in LDS ([128][C+8] bf16, 33-slot rows: conflict-free ds_read_b128) and
// consumed as two 64-column halves.
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

constexpr size_t kTileSmem = sizeof(TileLds) + 4 * 128 * sizeof(float2);

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct K1Workspace {
    float2* rowstat; float2* colpart; float2* colstat;
    float* rowbest_v; int* rowbest_j; float* colbest_part; int* match_j; int* counts; int* total;
    size_t bytes;
};

K1Workspace carve(void* ws, int Z, int L, int S) {
    K1Workspace w;
    int nI = (L + TILE_M - 1) / TILE_M;
    char* p = reinterpret_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off += align256(n); return q; };
    w.rowstat = (float2*)take((size_t)Z * L * sizeof(float2));
    w.colpart = (float2*)take((size_t)Z * nI * S * sizeof(float2));
    w.colstat = (float2*)take((size_t)Z * S * sizeof(float2));
    w.rowbest_v = (float*)take((size_t)Z * L * sizeof(float));
    w.rowbest_j = (int*)take((size_t)Z * L * sizeof(int));
    w.colbest_part = (float*)take((size_t)Z * nI * S * sizeof(float));
    w.match_j = (int*)take((size_t)Z * L * sizeof(int));
    w.counts = (int*)take((size_t)(Z + 1) * sizeof(int));
    w.total = w.counts ? w.counts + Z : nullptr;
    w.bytes = off;
    return w;
}

}  // namespace

extern "C" {

size_t far_dual_softmax_workspace_bytes(int Z, int L, int S) {
    return carve(nullptr, Z, L, S).bytes;
}

// Row/column softmax statistics of sim = ((f0/feat_div) . (f1/feat_div)) / sim_div * sim_mul.
// rowstat [Z][L] float2 (max, sum-exp); colstat [Z][S] float2.  ws from far_dual_softmax_workspace_bytes.
int far_dual_softmax_stats_f32(const float* f0, const float* f1, int Z, int L, int S, int C,
                               float feat_div, float sim_div, float sim_mul,
                               const uint8_t* mask0, const uint8_t* mask1,
                               float* rowstat_out, float* colstat_out, void* ws, hipStream_t stream) {
    far_clear_errors();
    if (!f0 || !f1 || !ws || Z <= 0 || L <= 0 || S <= 0 || C <= 0 || (C % KC) != 0) return FAR_EINVAL;
    K1Workspace w = carve(ws, Z, L, S);
    SimParams sp = make_sim(feat_div, sim_div, sim_mul);
    sp.stagger = far_get_tuning(0) & 1;
    int nI = (L + TILE_M - 1) / TILE_M;
    float2* rs = rowstat_out ? (float2*)rowstat_out : w.rowstat;
    float2* cs = colstat_out ? (float2*)colstat_out : w.colstat;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)k_stats_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileSmem);
        hipFuncSetAttribute((const void*)k_match_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileSmem);
        attr_set = true;
    }
    if (C == 64 && sp.feat_div == 1.0f && !mask0 && !mask1 && far_get_tuning(1) == 0) {
        const size_t smem = 128 * S64_KS * sizeof(float) + 4 * 64 * sizeof(float2);
        hipLaunchKernelGGL(k_stats_c64_f32, dim3(nI * Z), dim3(256), smem, stream, f0, f1, Z, L, S, sp, rs, w.colpart);
    } else {
        hipLaunchKernelGGL(k_stats_f32, dim3(nI * Z), dim3(256), kTileSmem, stream, f0, f1, Z, L, S, C, sp, mask0, mask1,
                           rs, w.colpart);
    }
    hipLaunchKernelGGL(k_colreduce, dim3((S + 255) / 256, Z), dim3(256), 0, stream, w.colpart, nI, S, cs);
    return far_check_launch();
}

// Full coarse matcher.  Outputs must hold Z*L entries (worst case); *total_out (device int) receives M.
// conf_out: optional [Z][L][S] (the reference's data['conf_matrix']).
// valid_hw: optional [Z][4] int (padded-mask datasets); scale0/scale1: optional [Z][2] float.
int far_coarse_match_f32(const float* f0, const float* f1, int Z, int L, int S, int C,
                         float temperature, float thr, int border, int h0, int w0, int h1, int w1,
                         float cell_scale, const uint8_t* mask0, const uint8_t* mask1,
                         const int* valid_hw, const float* scale0, const float* scale1,
                         float* conf_out, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                         float* mkpts0_c, float* mkpts1_c, int* counts_out, int* total_out,
                         void* ws, hipStream_t stream) {
    far_clear_errors();
    if (!f0 || !f1 || !ws || !b_ids || !i_ids || !j_ids || !mconf || !mkpts0_c || !mkpts1_c || !total_out)
        return FAR_EINVAL;
    if (Z <= 0 || L <= 0 || S <= 0 || C <= 0 || (C % KC) != 0 || h0 * w0 != L || h1 * w1 != S) return FAR_EINVAL;
    K1Workspace w = carve(ws, Z, L, S);
    int rc = far_dual_softmax_stats_f32(f0, f1, Z, L, S, C, sqrtf((float)C), temperature, 1.0f, mask0, mask1,
                                        nullptr, nullptr, ws, stream);
    if (rc) return rc;
    SimParams sp = make_sim(sqrtf((float)C), temperature, 1.0f);
    sp.stagger = (far_get_tuning(0) >> 1) & 1;
    int nI = (L + TILE_M - 1) / TILE_M;
    int* counts = counts_out ? counts_out : w.counts;
    hipMemsetAsync(counts, 0, sizeof(int) * Z, stream);
    hipLaunchKernelGGL(k_match_f32, dim3(nI * Z), dim3(256), kTileSmem, stream, f0, f1, Z, L, S, C, sp, mask0, mask1,
                       w.rowstat, w.colstat, conf_out, w.rowbest_v, w.rowbest_j, w.colbest_part);
    hipLaunchKernelGGL(k_finalize, dim3((L + 255) / 256, Z), dim3(256), 0, stream, w.rowbest_v, w.rowbest_j,
                       w.colbest_part, nI, L, S, thr, border, h0, w0, h1, w1, valid_hw, w.match_j, counts);
    hipLaunchKernelGGL(k_compact, dim3(Z), dim3(256), 0, stream, w.match_j, w.rowbest_v, counts, L, w0, w1,
                       cell_scale, scale0, scale1, b_ids, i_ids, j_ids, mconf, mkpts0_c, mkpts1_c, total_out);
    return far_check_launch();
}

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
    static bool attr_set = false;
    const size_t smem = 2 * 64 * (size_t)C * 2 + 2 * 4 * 64 * sizeof(float2);
    if (!attr_set) {
        hipFuncSetAttribute((const void*)k_stats_bf16<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute((const void*)k_match_bf16<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
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
