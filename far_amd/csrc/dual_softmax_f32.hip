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
#include "dual_softmax_common.h"

using namespace far_ds;

namespace {

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
        chunk_mfma(acc, lds, buf, wave, lane);
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
        chunk_mfma(acc, lds, buf, wave, lane);
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

constexpr size_t kTileSmem = sizeof(TileLds) + 4 * 128 * sizeof(float2);

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
    FAR_ONCE_PER_DEVICE(
        hipFuncSetAttribute((const void*)k_stats_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileSmem);
        hipFuncSetAttribute((const void*)k_match_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileSmem));
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
}  // extern "C"
