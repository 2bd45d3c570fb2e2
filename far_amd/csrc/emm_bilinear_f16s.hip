// K2 on the f16 matrix cores with split-precision operands: the bilinear dual-softmax attention of the EMM head.
//
// Same operator as emm_bilinear_f32.hip (reference: mp3d_loftr/src/loftr/loftr_module/transformer.py:275-292):
//   T = P [v | pos],  P = softmax(s, keys) * softmax(s, queries),  s = (q k^T) * scale          (Z, N, 70)
// but the two contractions run at the f16 matrix rate with fp32-grade operands: every fp32 value is split
// v = hi + lo (two fp16) and a product is hi.hi + hi.lo + lo.hi accumulated in fp32 (conv_igemm_f16s.hip explains
// the numerics).  The exact-f32 MFMA variant spends 64 cycles per 32x32x2 step; this one 3 x 32 cycles per 32x32x16.
//
// Launches of one call (all on the caller's stream, workspace from the caller):
//   k_prep_qk      q, k -> fp16 hi / lo planes, scaled by 2^4 (exact), written in the LDS image (16-byte slots
//                  XOR-swizzled per row) so that a 64-row tile is one linear LDS-DMA copy
//   k_rowstats     log2-domain softmax statistics (max, sum 2^(x - max)) of every COLUMN of s (the softmax over queries):
//                  run with (k, q), so that in the transposed score tile D[m = column][n = row] a lane owns one key --
//                  no cross-lane reductions.  (Round 2 ran it a second time with (q, k) for the softmax over keys;
//                  those statistics are now formed online inside k_pv: one score pass of three became two.)
//   k_prep_v       [v | pos] / colsum, transposed to [96 columns][keys] fp16 hi / lo tiles in LDS-image order, keys
//                  permuted inside each group of 32 into the k-order in which the accumulator registers hold P; the
//                  column reference C_j = ceil(colmax_j) as an integer, colsum rescaled to it
//   k_pv           per 128 queries: for every 64-key tile  scores (24 MFMAs) -> running integer row reference
//                  R_i = ceil(running row max) (a change rescales the accumulators by an exact power of two) ->
//                  e = 2^(x - R_i) (ONE exp per score), rowsum += e, p = e * ldexp(e, R_i - C_j + 15)
//                  = 2^(2 x - R_i - C_j + 15) <= 2^15 -> fp16 hi / lo straight from the accumulator registers ->
//                  T += P v~ (36 MFMAs); 1 / rowsum, 1 / colsum and the 2^-15 are folded into the output scale and v~
#include "common.h"
#include <type_traits>

typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int D = 64;            // head dim
constexpr int DV = 70;           // v~ width (64 + 6 positional)
constexpr int DVP = 96;          // padded to 3 MFMA column tiles
constexpr int KT = 64;           // keys per tile
// The tile's 64 integer column references ride in two padding rows of the v~^T hi plane (rows 70..95 only feed output
// columns nobody stores): 256 bytes at row CREF_ROW, written by k_prep_v, so they arrive in LDS with the tile's DMA.
constexpr int CREF_ROW = 80;
constexpr float QK_PRESCALE = 16.0f;     // q, k scaled by 2^4 before the split (their fp16 lo parts stay normal)
constexpr float V_PRESCALE = 64.0f;      // v~ / colsum scaled by 2^6; p by 2^15 (it is <= 1)
constexpr float NEG_HUGE = -1.0e30f;
constexpr int PAD_REF = 1 << 24;         // column reference of a padded key: p = ldexp(e, R - PAD_REF + 15) = 0
constexpr int ROW_REF0 = -(1 << 24);     // row reference before the first tile

// Where problem z = p * heads + hh of a [Z][N][64] operand lives: base + hh * head_stride + ((p + rot) % P) * prob_stride
// (floats; P = Z / heads).  Contiguous [Z][N][64]: heads = 1, prob_stride = N * 64.  The head's fused q | k | v
// projection writes per-(tensor, head) planes [2B][N][64]: heads = 4, head_stride = 2B N 64, prob_stride = N 64, and
// the query side of direction d is image 1 - d: rot = B.
struct ZLayout {
    int heads, P, rot;
    long head_stride, prob_stride;
    __device__ __forceinline__ size_t base(long z) const {
        const long pz = z / heads;
        const int hh = (int)(z - pz * heads);
        return (size_t)hh * head_stride + (size_t)((pz + rot) % P) * prob_stride;
    }
};

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// ---- k_prep_qk: x [Z][N][64] fp32 -> hi / lo [Z][Np][64] fp16 (rows >= N zero), 16-byte slot ^= (row >> 1) & 7
// overflow (device int or null): |= 1 when a value is beyond the range of the 2^4-scaled split (|x| > 4094: hi = inf)
__global__ void k_prep_qk(const float* __restrict__ x, ZLayout lay, int Z, int N, int Np, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                          int* __restrict__ overflow) {
    const long total = (long)Z * Np * 8;
    bool bad = false;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int slot = (int)(t & 7);
        const long row = t >> 3;
        const int i = (int)(row % Np);
        const long z = row / Np;
        f16x8 vh, vl;
        if (i < N) {
            const float* src = x + lay.base(z) + (size_t)i * D + slot * 8;
            const float4 a = *reinterpret_cast<const float4*>(src);
            const float4 b = *reinterpret_cast<const float4*>(src + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 h, l;
                split1(v[e] * QK_PRESCALE, h, l);
                vh[e] = h; vl[e] = l;
                bad |= !(fabsf(v[e]) <= 65504.0f / QK_PRESCALE);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { vh[e] = (_Float16)0.f; vl[e] = (_Float16)0.f; }
        }
        const int s2 = slot ^ ((i >> 1) & 7);
        *reinterpret_cast<f16x8*>(hi + (size_t)row * D + s2 * 8) = vh;
        *reinterpret_cast<f16x8*>(lo + (size_t)row * D + s2 * 8) = vl;
    }
    if (overflow && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(overflow, 1);
}

// Fragment of the register-resident ("row") side: row i, channels 16 s + 8 h .. + 7 of both planes (un-swizzled
// on the way in: `irow` is the row index inside its problem, which the slot XOR was derived from).
struct RowFrags {
    f16x8 hi[4], lo[4];
    __device__ __forceinline__ void load(const _Float16* ph, const _Float16* pl, size_t row, int irow, int h) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int slot = (2 * s + h) ^ ((irow >> 1) & 7);
            hi[s] = *reinterpret_cast<const f16x8*>(ph + row * D + 8 * slot);
            lo[s] = *reinterpret_cast<const f16x8*>(pl + row * D + 8 * slot);
        }
    }
    // Waits for the loads above HERE and tells the compiler so (the fragments are operands of the statement): left
    // "possibly pending", hipcc re-waits for them with a descending vmcnt ladder inside the tile loop, where the counter
    // belongs to the asm-issued tile requests -- each wait of the ladder would then stall on a tile still in flight.
    __device__ __forceinline__ void arrived() {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]),
                                            "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]) : : "memory");
    }
};

// One 64-row tile of the swizzled ("column") side in LDS: [plane][64 rows][128 B].  Linear DMA of 2 x 8 KiB.
constexpr int CT_PLANE = KT * 128;
__device__ __forceinline__ void dma_col_tile(unsigned char* lds, const _Float16* gh, const _Float16* gl, size_t row0, int tid, int wave) {
    const unsigned char* sh = reinterpret_cast<const unsigned char*>(gh + row0 * D) + tid * 16;
    const unsigned char* sl = reinterpret_cast<const unsigned char*>(gl + row0 * D) + tid * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_global_load_lds((gptr_t)(sh + j * 4096), (lptr_t)(lds + j * 4096 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(sl + j * 4096), (lptr_t)(lds + CT_PLANE + j * 4096 + wave * 1024), 16, 0, 0);
    }
}

// The same requests as `asm`: once a kernel keeps a tile in flight WHILE it computes on the previous one, hipcc's wait
// insertion puts s_waitcnt vmcnt(0) in front of the first ds_read that follows an LDS-DMA builtin (it cannot tell the
// two stages apart), which serialises load and compute again.  Requests issued from asm are invisible to that
// bookkeeping; completion is waited for by hand at the top of the loop, visibility to the other waves comes from the
// barrier.  M0 (LDS base of the request) is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_uniform(const unsigned char* p) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)p);
}
__device__ __forceinline__ void dma_col_tile_async(unsigned char* lds, const _Float16* gh, const _Float16* gl, size_t row0, int tid, int wave) {
    const unsigned char* sh = reinterpret_cast<const unsigned char*>(gh + row0 * D) + tid * 16;
    const unsigned char* sl = reinterpret_cast<const unsigned char*>(gl + row0 * D) + tid * 16;
    const unsigned dst = lds_addr_uniform(lds + wave * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        glds16(sh + j * 4096, dst + j * 4096);
        glds16(sl + j * 4096, dst + CT_PLANE + j * 4096);
    }
}

// acc[ct] (32 tile columns x this wave's 32 rows, transposed: D[m = tile row 32 ct + ..][n = row i]) = x(col) . x(row)
// SPLIT = false (round 5, far_emm_pv_f16): plain fp16 operands -- hi.hi only -- the 16-bit-operand precision class of BASELINE configs[1]
template <bool SPLIT = true>
__device__ __forceinline__ void score_tile(f32x16 (&acc)[2], const unsigned char* lds, const RowFrags& rf, int l31, int h) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f16x8 ch[2], cl[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int row = 32 * ct + l31;
            const int off = row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) * 16);
            ch[ct] = *reinterpret_cast<const f16x8*>(lds + off);
            if (SPLIT) cl[ct] = *reinterpret_cast<const f16x8*>(lds + CT_PLANE + off);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[ct], rf.hi[s], acc[ct], 0, 0, 0);
        if (SPLIT) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[ct], rf.lo[s], acc[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl[ct], rf.hi[s], acc[ct], 0, 0, 0);
        }
    }
}

// ---- k_rowstats: stat[z][i] = (max_j x_ij, sum_j 2^(x_ij - max)),  x = (a_i . b_j) * c1   (log2 domain)
// With 64-channel heads a tile is 24 MFMAs (768 matrix-pipe cycles at the nominal clock) against ~135 softmax VALU
// instructions (33 of them exps at quarter rate).  Measured on 256 problems of 4800 x 4800 (round 2, rocprofv3): the MFMAs
// and tile traffic alone take 1.83 ms, the VALU work alone 1.12 ms, the kernel 2.35 ms -- i.e. the matrix pipe under this
// load (three dense f16 MFMAs per product, clock power-limited to ~1.3-1.5 GHz) is the bound, and what is left to gain is
// the 0.5 ms of imperfect overlap.  Tried one by one, each within 5 % of 2.4 ms: padding test behind a wave-uniform branch +
// packed fp32 ops (a third fewer VALU instructions), double-buffered tiles through asm LDS-DMA, two row tiles per wave
// (half the LDS fragment reads per MFMA), and the in-wave software pipeline kept here: the MFMAs of tile t+1 interleaved
// (sched_group_barrier) with the softmax update of tile t, whose scores sit in a second accumulator pair.
template <bool MASKED>
__device__ __forceinline__ void rowstat_update(f32x16 (&acc)[2], float c1, int j0, int N, int h, float& m, float& sum, float& comp) {
    float tm = NEG_HUGE;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = acc[ct][r] * c1;
            if (MASKED && j0 + 32 * ct + mfma32_row(r, h) >= N) x = NEG_HUGE;
            acc[ct][r] = x;
            tm = fmaxf(tm, x);
        }
    // The tile's 32 terms are summed on their own and then added to the running sum with Kahan compensation: once the
    // row maximum (a term equal to 1) is in the accumulator, the other terms (~1e-8 each for a confident match) are
    // below half an ulp of it and a plain fp32 running sum would drop them one by one (the "swamping" that costs the
    // fp32 reference ~7e-5 on conf, oracle/coarse.py).  Pairs of scores go through the packed fp32 pipe.
    const float mn = fmaxf(m, tm);
    const float resc = __builtin_amdgcn_exp2f(m - mn);
    f32x2 t2 = {0.f, 0.f};
    const f32x2 mn2 = {mn, mn};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 d = f32x2{acc[ct][r], acc[ct][r + 1]} - mn2;
            t2 += f32x2{__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
        }
    const float t = t2.x + t2.y;
    sum *= resc;
    comp *= resc;
    const float y = t - comp;
    const float ns = sum + y;
    comp = (ns - sum) - y;
    sum = ns;
    m = mn;
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_rowstats(const _Float16* __restrict__ ah, const _Float16* __restrict__ al,
                                                     const _Float16* __restrict__ bh, const _Float16* __restrict__ bl,
                                                     int Z, int N, int Np, float c1, float2* __restrict__ stat) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[2 * 2 * CT_PLANE];      // two stages
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    int z, Ib;
    tile_coords(Np / 128, Z, z, Ib);
    const int irow = Ib * 128 + 32 * wave + l31;
    RowFrags rf;
    rf.load(ah, al, (size_t)z * Np + irow, irow, h);
    rf.arrived();
    float m = NEG_HUGE, sum = 0.f, comp = 0.f;
    const int ntile = Np / KT;
    auto request = [&](int jt) { dma_col_tile_async(lds_all + (jt & 1) * (2 * CT_PLANE), bh, bl, (size_t)z * Np + jt * KT, tid, wave); };
    request(0);
    if (ntile > 1) request(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 cur[2];
    score_tile<SPLIT>(cur, lds_all, rf, l31, h);
    for (int jt = 0; jt + 1 < ntile; ++jt) {
        // cur = scores of tile jt.  Tile jt + 1 (requested an iteration ago) has landed; after the barrier nobody reads
        // tile jt's stage any more (its scores were formed before the previous barrier): tile jt + 2 goes there.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (jt + 2 < ntile) request(jt + 2);
        const unsigned char* lds = lds_all + ((jt + 1) & 1) * (2 * CT_PLANE);
        f32x16 nxt[2];
        if ((jt + 1) * KT <= N) {                                     // wave-uniform: tile jt entirely inside the sequence
            score_tile<SPLIT>(nxt, lds, rf, l31, h);
            rowstat_update<false>(cur, c1, jt * KT, N, h, m, sum, comp);
#pragma unroll
            for (int i = 0; i < (SPLIT ? 24 : 0); ++i) {
                if (i < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // a fragment read (16 per tile), ahead of its MFMA
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA of tile jt + 1
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);               // softmax VALU of tile jt (~135 per tile, 33 of them exps)
            }
        } else {
            score_tile<SPLIT>(nxt, lds, rf, l31, h);
            rowstat_update<true>(cur, c1, jt * KT, N, h, m, sum, comp);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) cur[ct] = nxt[ct];
    }
    rowstat_update<true>(cur, c1, (ntile - 1) * KT, N, h, m, sum, comp);
    sum -= comp;
    // the two half-waves hold the two key halves of the same rows
    const float mo = shfl_xor_f(m, 32), so = shfl_xor_f(sum, 32);
    const float mn = fmaxf(m, mo);
    const float st = sum * __builtin_amdgcn_exp2f(m - mn) + so * __builtin_amdgcn_exp2f(mo - mn);
    if (h == 0 && irow < N) stat[(size_t)z * N + irow] = make_float2(mn, st);
}

// ---- k_prep_v: v~^T / colsum in LDS-image tiles: out[plane][z][jt][96 b][8 slots ^ ((b >> 1) & 7)][8 keys (permuted)]
// position p = 8 g + e of a 32-key group holds key  16 (g >> 1) + 4 (g & 1) + (e < 4 ? e : e + 4)
__global__ __launch_bounds__(256) void k_prep_v(const float* __restrict__ v, ZLayout lay, const float* __restrict__ pos,
                                               const float2* __restrict__ colstat, int Z, int N, int Np,
                                               _Float16* __restrict__ oh, _Float16* __restrict__ ol,
                                               int* __restrict__ cref) {
    __shared__ float tile[KT][DV + 3];
    const int ntile = Np / KT;
    const int z = blockIdx.x / ntile, jt = blockIdx.x - z * ntile;
    const int tid = threadIdx.x;
    for (int i = tid; i < KT * DV; i += 256) {
        const int r = i / DV, c = i - r * DV;
        const int j = jt * KT + r;
        float x = 0.f;
        if (j < N) {
            x = c < D ? v[lay.base(z) + (size_t)j * D + c] : pos[(size_t)j * 6 + (c - D)];
            // the column's statistics re-based from its maximum m to the integer reference C = ceil(m):
            // sum 2^(x - C) = colsum 2^(m - C)
            const float2 cs = colstat[(size_t)z * N + j];
            x = x / (cs.y * __builtin_amdgcn_exp2f(cs.x - ceilf(cs.x))) * V_PRESCALE;
        }
        tile[r][c] = x;
    }
    if (tid < KT) {                                      // integer column references; keys past N get +huge -> p = 0
        const int j = jt * KT + tid;
        cref[(size_t)z * Np + j] = j < N ? (int)ceilf(fminf(fmaxf(colstat[(size_t)z * N + j].x, -1.0e6f), 1.0e6f)) : PAD_REF;
    }
    __syncthreads();
    const size_t base = ((size_t)z * ntile + jt) * DVP * KT;
    for (int i = tid; i < DVP * 8; i += 256) {
        const int b = i >> 3, slot = i & 7;                // slot = 4 ct + g
        const int ct = slot >> 2, g = slot & 3;
        f16x8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = 32 * ct + 16 * (g >> 1) + 4 * (g & 1) + (e < 4 ? e : e + 4);
            const float x = b < DV ? tile[key][b] : 0.f;
            _Float16 hh, ll;
            split1(x, hh, ll);
            vh[e] = hh; vl[e] = ll;
        }
        const size_t o = base + (size_t)b * KT + ((slot ^ ((b >> 1) & 7)) * 8);
        *reinterpret_cast<f16x8*>(oh + o) = vh;
        *reinterpret_cast<f16x8*>(ol + o) = vl;
    }
    __syncthreads();                                     // the zero rows are written: now the references over two of them
    if (tid < KT) reinterpret_cast<int*>(oh + base + (size_t)CREF_ROW * KT)[tid] = cref[(size_t)z * Np + jt * KT + tid];
}

// ---- k_pv: T[z][i][:] = (2^-15 / rowsum_i) * sum_j 2^(2 x_ij - R_i - C_j + 15) * (v~_j / colsum_j),  rowsum_i = sum_j 2^(x_ij - R_i)
// with the row statistics formed ONLINE: R_i = ceil(max_j x_ij) over the keys seen so far, an integer, so that a change of
// reference rescales the running sum and the T accumulators by an exact power of two (v_ldexp), and so that the one
// exponential per score e = 2^(x - R_i) serves both softmaxes: the row sum takes e, the product takes
// p = e * ldexp(e, R_i - C_j + 15) (x <= R_i and x <= C_j: p <= 2^15, e <= 1; padded keys have C_j = 2^24: p = 0).
// Both half-waves hold keys of the same 32 rows and feed ONE accumulator through the MFMA's k dimension, so they agree on
// R_i (one cross-half exchange per tile).  Writes rowstat[z][i] = (R_i, rowsum_i) for the backward kernels.
constexpr int VT_PLANE = DVP * 128;     // one v~^T tile plane: 96 rows x 64 keys fp16
constexpr int PV_STAGE = 2 * CT_PLANE + 2 * VT_PLANE;     // 40 KiB: two stages = half a CU's LDS, two workgroups per CU
template <bool MASKED>
__device__ __forceinline__ float tile_rowmax(const f32x16 (&acc)[2], float c1, int j0, int N, int h) {
    float tm = NEG_HUGE;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool pad = MASKED && j0 + 32 * ct + mfma32_row(r, h) >= N;
            tm = fmaxf(tm, pad ? NEG_HUGE : acc[ct][r]);
        }
    return tm * c1;                       // c1 > 0
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_pv(const _Float16* __restrict__ qh, const _Float16* __restrict__ ql,
                                               const _Float16* __restrict__ kh, const _Float16* __restrict__ kl,
                                               const _Float16* __restrict__ vh, const _Float16* __restrict__ vl,
                                               const int* __restrict__ cref, int Z, int N, int Np, float c1,
                                               float2* __restrict__ rowstat, float* __restrict__ T) {
    // two stages of { key tile (hi, lo), v~^T tile (hi, lo) with the tile's column references in two of its padding rows }:
    // tile jt + 1 travels while tile jt is computed (asm LDS-DMA, one barrier per tile)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    int z, Ib;
    tile_coords(Np / 128, Z, z, Ib);
    const int i0 = Ib * 128 + 32 * wave;
    const int irow = i0 + l31;
    RowFrags rf;
    rf.load(qh, ql, (size_t)z * Np + irow, irow, h);
    rf.arrived();

    f32x16 tacc[3];
#pragma unroll
    for (int bt = 0; bt < 3; ++bt)
#pragma unroll
        for (int r = 0; r < 16; ++r) tacc[bt][r] = 0.f;
    int R = ROW_REF0;                      // integer row reference, equal in lanes l and l + 32
    float sum = 0.f, comp = 0.f;           // this half-wave's share of rowsum (Kahan-compensated per tile)

    const int ntile = Np / KT;
    const unsigned char* vsrc_h = reinterpret_cast<const unsigned char*>(vh + (size_t)z * ntile * DVP * KT) + tid * 16;
    const unsigned char* vsrc_l = reinterpret_cast<const unsigned char*>(vl + (size_t)z * ntile * DVP * KT) + tid * 16;
    auto request = [&](int jt) {
        unsigned char* const st = lds_all + (jt & 1) * PV_STAGE;
        dma_col_tile_async(st, kh, kl, (size_t)z * Np + jt * KT, tid, wave);
        const unsigned dv = lds_addr_uniform(st + 2 * CT_PLANE + wave * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            glds16(vsrc_h + (size_t)jt * VT_PLANE + j * 4096, dv + j * 4096);
            glds16(vsrc_l + (size_t)jt * VT_PLANE + j * 4096, dv + VT_PLANE + j * 4096);
        }
    };
    request(0);
    for (int jt = 0; jt < ntile; ++jt) {
        // tile jt (requested an iteration ago) has landed; after the barrier nobody reads the other stage any more
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (jt + 1 < ntile) request(jt + 1);
        const unsigned char* const lds = lds_all + (jt & 1) * PV_STAGE;
        const unsigned char* const ldv = lds + 2 * CT_PLANE;
        const int* const lcr = reinterpret_cast<const int*>(ldv + CREF_ROW * 128);

        f32x16 acc[2];
        score_tile<SPLIT>(acc, lds, rf, l31, h);
        // ---- row reference: ceil of the running maximum, agreed between the two half-waves
        const bool ragged = (jt + 1) * KT > N;                                   // wave-uniform
        const float tm = ragged ? tile_rowmax<true>(acc, c1, jt * KT, N, h) : tile_rowmax<false>(acc, c1, jt * KT, N, h);
        int Rn = (int)ceilf(fminf(fmaxf(tm, -1.0e6f), 1.0e6f));
        Rn = Rn > R ? Rn : R;
        const int Ro = __shfl_xor(Rn, 32, 64);
        Rn = Rn > Ro ? Rn : Ro;
        if (__any(Rn != R)) {                                                    // rare after the first tiles
            int d = R - Rn;                                                      // <= 0
            d = d < -200 ? -200 : d;                                             // 2^-200 flushes every fp32 to zero already
            sum = ldexpf(sum, d);
            comp = ldexpf(comp, d);
#pragma unroll
            for (int r = 0; r < 16; ++r) {                                       // accumulator register r holds row mfma32_row(r, h)
                const int dr = __shfl(d, mfma32_row(r, h), 64);
#pragma unroll
                for (int bt = 0; bt < 3; ++bt) tacc[bt][r] = ldexpf(tacc[bt][r], dr);
            }
            R = Rn;
        }
        const float nR = -(float)R;
        f32x2 t2 = {0.f, 0.f};
        // the ragged-tile test is hoisted out of the loop (two straight-line copies of it): inside, the compiler split the
        // tile into four branchy blocks of 9 MFMAs that could not overlap each other's exponentials
        auto pv_tile = [&](auto ragged_c) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // p for this lane's 8 keys of MFMA (ct, u): accumulator registers 8 u .. 8 u + 7
                f16x8 ph, pl;
                // column references of this lane's keys: j = jt*64 + 32 ct + 8 (r >> 2) + 4 h + (r & 3), r = 8 u .. 8 u + 7
                const int4 cA = *reinterpret_cast<const int4*>(lcr + 32 * ct + 16 * u + 4 * h);
                const int4 cB = *reinterpret_cast<const int4*>(lcr + 32 * ct + 16 * u + 8 + 4 * h);
                const int cmv[8] = {cA.x, cA.y, cA.z, cA.w, cB.x, cB.y, cB.z, cB.w};
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const int r = 8 * u + e;
                    const f32x2 x2 = __builtin_elementwise_fma(f32x2{acc[ct][r], acc[ct][r + 1]}, f32x2{c1, c1}, f32x2{nR, nR});
                    float e0 = __builtin_amdgcn_exp2f(x2.x);
                    float e1 = __builtin_amdgcn_exp2f(x2.y);
                    if (RAGGED) {                                                // zero-padded keys score 0, not -inf
                        if (jt * KT + 32 * ct + mfma32_row(r, h) >= N) e0 = 0.f;
                        if (jt * KT + 32 * ct + mfma32_row(r + 1, h) >= N) e1 = 0.f;
                    }
                    t2 += f32x2{e0, e1};
                    const f32x2 p2 = f32x2{e0, e1} * f32x2{ldexpf(e0, R - cmv[e] + 15), ldexpf(e1, R - cmv[e + 1] + 15)};
                    f16x2 h2, l2;
                    if (SPLIT) split2(p2, h2, l2);
                    else { h2 = __builtin_convertvector(p2, f16x2); l2 = h2; }
                    ph[e] = h2.x; ph[e + 1] = h2.y;
                    pl[e] = l2.x; pl[e + 1] = l2.y;
                }
                const int slot = 4 * ct + 2 * u + h;
#pragma unroll
                for (int bt = 0; bt < 3; ++bt) {
                    const int b = 32 * bt + l31;
                    const int off = b * 128 + ((slot ^ ((b >> 1) & 7)) * 16);
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(ldv + off);
                    tacc[bt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, bh, tacc[bt], 0, 0, 0);
                    if (SPLIT) {
                        const f16x8 bl = *reinterpret_cast<const f16x8*>(ldv + VT_PLANE + off);
                        tacc[bt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, bl, tacc[bt], 0, 0, 0);
                        tacc[bt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, bh, tacc[bt], 0, 0, 0);
                    }
                }
            }
        }
        };
        if (ragged) pv_tile(std::true_type{});
        else pv_tile(std::false_type{});
        // the tile's 32 terms summed on their own, then added with Kahan compensation (see rowstat_update)
        const float y = (t2.x + t2.y) - comp;
        const float ns = sum + y;
        comp = (ns - sum) - y;
        sum = ns;
    }
    sum -= comp;
    const float rsum = sum + shfl_xor_f(sum, 32);          // both halves carry the same reference R
    if (h == 0 && irow < N) rowstat[(size_t)z * N + irow] = make_float2((float)R, rsum);
    // ---- store: lane holds column b = 32 bt + l31 of rows i0 + mfma32_row(r, h); the row scale comes from lane (row)
    const float oscale = (3.0517578125e-05f / V_PRESCALE) / rsum;       // 2^-15 2^-6 / rowsum of this lane's row (= l31)
#pragma unroll
    for (int bt = 0; bt < 3; ++bt) {
        const int b = 32 * bt + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = mfma32_row(r, h);
            const float sc = __shfl(oscale, rr, 64);
            const int i = i0 + rr;
            if (b < DV && i < N) T[((size_t)z * N + i) * DV + b] = tacc[bt][r] * sc;
        }
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct EmmWs {
    _Float16 *qh, *ql, *kh, *kl, *vh, *vl;
    float2 *rowstat, *colstat;
    int* cref;
    size_t bytes;
};
inline EmmWs carve(void* base, int Z, int N) {
    const int Np = (N + 127) / 128 * 128;
    unsigned char* p = (unsigned char*)base;
    size_t o = 0;
    auto take = [&](size_t n) { unsigned char* r = p ? p + o : nullptr; o += align256(n); return r; };
    EmmWs w;
    const size_t qk = (size_t)Z * Np * D * 2, vt = (size_t)Z * (Np / KT) * DVP * KT * 2, st = (size_t)Z * N * 8;
    w.qh = (_Float16*)take(qk); w.ql = (_Float16*)take(qk); w.kh = (_Float16*)take(qk); w.kl = (_Float16*)take(qk);
    w.vh = (_Float16*)take(vt); w.vl = (_Float16*)take(vt);
    w.rowstat = (float2*)take(st); w.colstat = (float2*)take(st);
    w.cref = (int*)take((size_t)Z * Np * 4);
    w.bytes = o;
    return w;
}

}  // namespace

extern "C" {

size_t far_emm_pv_f16s_workspace_bytes(int Z, int N) {
    if (Z <= 0 || N <= 0) return 0;
    return carve(nullptr, Z, N).bytes;
}

}  // extern "C"

template <bool SPLIT>
static int emm_pv_launch(const float* q, const float* k, const float* v, const float* pos, int Z, int N, int Dh, float scale,
                         int heads, long head_stride, long prob_stride, int q_rot, void* ws, float* T_out, int* overflow,
                         hipStream_t stream) {
    far_clear_errors();
    if (!q || !k || !v || !pos || !ws || !T_out || Z <= 0 || N <= 0 || Dh != D || heads < 1 || Z % heads || q_rot < 0 ||
        q_rot >= Z / heads)
        return FAR_EINVAL;
    const ZLayout lay{heads, Z / heads, 0, head_stride, prob_stride};
    const ZLayout layq{heads, Z / heads, q_rot, head_stride, prob_stride};
    const int Np = (N + 127) / 128 * 128;
    const EmmWs w = carve(ws, Z, N);
    const float c1 = scale * 1.44269504088896341f / (QK_PRESCALE * QK_PRESCALE);      // scores -> log2 domain
    const unsigned gprep = (unsigned)(((long)Z * Np * 8 + 255) / 256 < 65536L * 4 ? ((long)Z * Np * 8 + 255) / 256 : 65536L * 4);
    hipLaunchKernelGGL(k_prep_qk, dim3(gprep), dim3(256), 0, stream, q, layq, Z, N, Np, w.qh, w.ql, overflow);
    hipLaunchKernelGGL(k_prep_qk, dim3(gprep), dim3(256), 0, stream, k, lay, Z, N, Np, w.kh, w.kl, overflow);
    const dim3 grid((unsigned)(Np / 128) * Z);
    // softmax over queries (rows of the statistics kernel = keys); the softmax over keys is formed online inside k_pv
    const dim3 gstat((unsigned)(Np / 128) * Z);
    hipLaunchKernelGGL(k_rowstats<SPLIT>, gstat, dim3(256), 0, stream, w.kh, w.kl, w.qh, w.ql, Z, N, Np, c1, w.colstat);
    hipLaunchKernelGGL(k_prep_v, dim3((unsigned)(Np / KT) * Z), dim3(256), 0, stream, v, lay, pos, w.colstat, Z, N, Np, w.vh, w.vl, w.cref);
    bool cfg_failed = false;
    FAR_ONCE_PER_DEVICE(cfg_failed = hipFuncSetAttribute((const void*)k_pv<SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PV_STAGE) != hipSuccess);
    if (cfg_failed) return far_check_launch();
    hipLaunchKernelGGL(k_pv<SPLIT>, grid, dim3(256), 2 * PV_STAGE, stream, w.qh, w.ql, w.kh, w.kl, w.vh, w.vl, w.cref, Z, N, Np, c1,
                       w.rowstat, T_out);
    return far_check_launch();
}

extern "C" {

// T[z] = P[z] @ [v[z] | pos], P = softmax over keys * softmax over queries of s = (q k^T) * scale  -- the whole K2
// operator (statistics included) on split-fp16 operands.  pos [N][6], T_out [Z][N][70] fp32; q, k, v: problem
// z = p * heads + hh starts at  ptr + hh * head_stride + p' * prob_stride  floats and is [N][64] contiguous, with
// p' = p for k, v and (p + q_rot) mod (Z / heads) for q  (contiguous [Z][N][64]: heads = 1, prob_stride = 64 N, q_rot = 0);
// ws: far_emm_pv_f16s_workspace_bytes(Z, N) bytes.  overflow: device int or NULL, |= 1 when a q / k value is beyond the range
// of the 2^4-scaled fp16 split (|x| > 4094): T is then inf / NaN (far_emm_pv_f32 has no such limit).
// (v~ / colsum is at most 64 |v| in the split's units: |v| > 1023 would overflow it too; the head's v is a Linear of a
// LayerNorm output, the caller's range check covers it through the projection's own K9 flag.)
int far_emm_pv_f16s(const float* q, const float* k, const float* v, const float* pos, int Z, int N, int Dh, float scale,
                    int heads, long head_stride, long prob_stride, int q_rot, void* ws, float* T_out, int* overflow,
                    hipStream_t stream) {
    return emm_pv_launch<true>(q, k, v, pos, Z, N, Dh, scale, heads, head_stride, prob_stride, q_rot, ws, T_out, overflow, stream);
}

// The 16-bit-operand form (round 5): the same kernels with plain fp16 operands (one MFMA product per tile instead of the
// three of the split form; fp32 accumulation and the fp32 softmax statistics unchanged).  Same arguments, workspace and
// overflow flag as far_emm_pv_f16s; results agree with it to ~1e-3 relative (tests/test_emm_gpu.py).
int far_emm_pv_f16(const float* q, const float* k, const float* v, const float* pos, int Z, int N, int Dh, float scale,
                   int heads, long head_stride, long prob_stride, int q_rot, void* ws, float* T_out, int* overflow,
                   hipStream_t stream) {
    return emm_pv_launch<false>(q, k, v, pos, Z, N, Dh, scale, heads, head_stride, prob_stride, q_rot, ws, T_out, overflow, stream);
}

// The softmax statistics far_emm_pv_f16s left in its workspace, for the backward kernels (emm_bilinear_bwd_f16.hip):
// rowstat[z][i] = (ref, sum 2^(x - ref)) over keys of query row i with ref = ceil(max) (any reference >= the maximum
// gives the same softmax), colstat[z][j] = (max, sum) over queries of key column j (log2 domain).
int far_emm_pv_f16s_copy_stats(const void* ws, int Z, int N, float* rowstat_out, float* colstat_out, hipStream_t stream) {
    far_clear_errors();
    if (!ws || !rowstat_out || !colstat_out || Z <= 0 || N <= 0) return FAR_EINVAL;
    const EmmWs w = carve(const_cast<void*>(ws), Z, N);
    hipMemcpyAsync(rowstat_out, w.rowstat, (size_t)Z * N * 8, hipMemcpyDeviceToDevice, stream);
    hipMemcpyAsync(colstat_out, w.colstat, (size_t)Z * N * 8, hipMemcpyDeviceToDevice, stream);
    return far_check_launch();
}

}  // extern "C"
