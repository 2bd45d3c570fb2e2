// K1, materialising mode: data['conf_matrix'] (Z, L, S) fp32 at HBM write speed.
//
// Replaces coarse_matching.py:108-118 when the dense matrix itself is wanted (the reference's dense loss / plotting,
// loftr_loss.py:307-311).  The fused matcher (dual_softmax_f16s.hip) is MFMA-bound at three f16 MFMAs per product and
// wrote the matrix at 0.7 TB/s; here the 3.26 GB of a 32-pair batch are the bound, so the arithmetic is arranged
// around the stores:
//   * statistics (row / column max and sum) come from the split-precision passes of dual_softmax_f16s.hip: fp32-grade;
//   * k1_conf       one v_mfma_f32_32x32x16_f16 per 16 channels on the fp16 `hi` planes only (x' = fp16-operand
//                   score), p' = 2^(2 x' - rowmax - colmax) / (rowsum colsum): one exp per score.  The tile is NOT
//                   transposed: a lane owns one COLUMN, so that each store instruction of a wave writes two whole
//                   128-byte lines of two rows (dword stores, fully coalesced), with no LDS staging;
//                   column tiles arrive by LDS-DMA into a double buffer, one barrier per tile, the stores of tile t
//                   drain under the MFMAs of tile t + 1;  work item = (pair, 128-row block, column chunk), so that
//                   the grid is ~12 rounds of the chip (no tail) and all items of a pair run on one XCD (L2-resident
//                   operand planes);
//   * exactness     |p' - p| <= 2 ln2 |x' - x| p: the fp16 operand error (|x' - x| ~ 5e-4 in log2 units, rigorous
//                   bound 2^-10 sum|a_c b_c| log2e / (C temperature)) only matters where p is not tiny.  The second
//                   statistics pass (k1_rowstats<CAND>, dual_softmax_f16s.hip) lists every entry whose row softmax is
//                   >= 2^-12 -- a superset of p >= 2^-12, about one entry per row -- WITH its split-precision score x;
//                   k1_conf_fix rewrites those with the fused matcher's formula on that x.  It has to be that x: the
//                   statistics were accumulated from it, and an independently computed score, however exact, would not
//                   cancel against them (a float64 dot product here measured 1.6e-5 on confident entries; the fp32
//                   accumulation error of a 256-term dot is ~1e-5 in the log2 domain).  Every unlisted entry is below
//                   2^-12, where the fp16-operand error is < 3e-6 absolute: conf_matrix is within 1e-5 of the float64
//                   oracle everywhere (tests/test_coarse_gpu.py), although 99.99 % of it never saw an fp32-grade product.
#include "dual_softmax_common.h"

namespace far_conf {

using namespace far_ds;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int C = 256;
constexpr int NS = C / 16;
constexpr int KT = 64;                 // columns per tile
constexpr int ROWB = C * 2;            // bytes per fp16 row
constexpr int TILE = KT * ROWB;        // 32 KiB
constexpr int CHUNK_MAX = 16;          // column tiles per work item (upper bound: sizes the LDS copy of the column statistics)

// one 64-column tile of the `hi` plane (LDS image: 16-byte slot ^= row & 15, written by k1_prep) -> LDS, 8 KiB per wave
__device__ __forceinline__ void dma_tile(unsigned char* lds, const _Float16* g, size_t row0, int tid, int wave) {
    const unsigned char* s = reinterpret_cast<const unsigned char*>(g + row0 * C) + tid * 16;
#pragma unroll
    for (int j = 0; j < TILE / 4096; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(s + j * 4096), (lptr_t)(lds + j * 4096 + wave * 1024), 16, 0, 0);
}

// grid: Z * nI * NCH work items; item -> (z, row block Ib, column chunk ch); all items of one z on one XCD
template <bool MASKS>
__global__ __launch_bounds__(256, 2) void k1_conf(const _Float16* __restrict__ ah, const _Float16* __restrict__ bh,
                                                  int Z, int L, int S, int Lp, int Sp, int nI, int nch, int tiles_per_chunk,
                                                  float c2, float fill2x2, const uint8_t* __restrict__ mask0,
                                                  const uint8_t* __restrict__ mask1, const float2* __restrict__ rowstat,
                                                  const float* __restrict__ cmax, const float* __restrict__ cinv,
                                                  float* __restrict__ conf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    int z, item;
    tile_coords(nI * nch, Z, z, item);
    const int Ib = item / nch, ch = item - Ib * nch;
    const int ntile = (S + KT - 1) / KT;
    const int t0 = ch * tiles_per_chunk, t1 = min(ntile, t0 + tiles_per_chunk);
    if (t0 >= t1) return;                                        // whole workgroup (uniform)
    const int row0 = Ib * 128 + 32 * wave;                       // first row of this wave
    // A operand: this wave's 32 rows, all 256 channels, in registers (64 VGPRs); lane = (row l31, k-half h)
    f16x8 af[NS];
    {
        const int irow = row0 + l31;
        const _Float16* p = ah + ((size_t)z * Lp + irow) * C;
#pragma unroll
        for (int s = 0; s < NS; ++s) af[s] = *reinterpret_cast<const f16x8*>(p + 8 * ((2 * s + h) ^ (irow & 15)));
    }
    // log2 conf = 2 x - (rowmax + log2 rowsum) - (colmax + log2 colsum): one fma + one exp per entry.
    // Row terms of the 16 rows whose results this lane holds: i = row0 + (r & 3) + 8 (r >> 2) + 4 h
    float rl[16];
    unsigned rmask_bits = 0;                                     // bit r: row masked out (mask0 == 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = row0 + mfma32_row(r, h);
        const float2 st = i < L ? rowstat[(size_t)z * L + i] : make_float2(0.f, 1.f);
        rl[r] = st.x + __builtin_amdgcn_logf(st.y);
        if (MASKS && mask0 && i < L && !mask0[(size_t)z * L + i]) rmask_bits |= 1u << r;
    }
    const bool rows_live = row0 < L;                             // wave-uniform: any valid row in this wave
    const bool rows_full = row0 + 32 <= L;                       // wave-uniform: all 32 rows valid
    // stores: lane (l31, h) holds, for column j = jt*64 + 32 ct + l31, the rows row0 + (r & 3) + 8 (r >> 2) + 4 h, so one
    // store instruction (register r of all lanes) writes 32 consecutive floats of two rows: two whole 128-byte lines.
    // Address = wave-uniform base of row (r & 3) + 8 (r >> 2) [SGPR pair] + this lane's 32-bit byte offset + immediate.
    unsigned char* const crow = reinterpret_cast<unsigned char*>(conf + ((size_t)z * L + row0) * S);
    const unsigned lane_off = (unsigned)(4 * h * S + l31) * 4u;
    const size_t row_bytes = (size_t)S * 4;
    // this chunk's column terms -> LDS (read back with LDS latency in every tile's epilogue; padded columns: +inf -> p = 0)
    float* const cstat = reinterpret_cast<float*>(lds + 2 * TILE);           // [CHUNK_MAX * KT]
    for (int o = tid; o < (t1 - t0) * KT; o += 256)
        cstat[o] = cmax[(size_t)z * Sp + t0 * KT + o] - __builtin_amdgcn_logf(cinv[(size_t)z * Sp + t0 * KT + o]);
    dma_tile(lds, bh, (size_t)z * Sp + (size_t)t0 * KT, tid, wave);
    // The stores of a tile are issued one tile LATE (after the next tile's barrier), so that the vmcnt(0) each barrier needs
    // for the LDS-DMA never waits for stores younger than a whole tile of MFMA work: `hold` carries the tile across.
    f32x16 hold[2];
    auto store_tile = [&](int jt) {
        unsigned char* const tbase = crow + (size_t)jt * KT * 4;
        if (rows_full && (jt + 1) * KT <= S) {                              // wave-uniform: no per-store predicate
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float*>(tbase + (size_t)((r & 3) + 8 * (r >> 2)) * row_bytes + 128 * ct + lane_off) = hold[ct][r];
        } else if (rows_live) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int j = jt * KT + 32 * ct + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (j < S && row0 + mfma32_row(r, h) < L)
                        *reinterpret_cast<float*>(tbase + (size_t)((r & 3) + 8 * (r >> 2)) * row_bytes + 128 * ct + lane_off) = hold[ct][r];
            }
        }
    };
    for (int jt = t0; jt < t1; ++jt) {
        unsigned char* cur = lds + ((jt - t0) & 1) * TILE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // tile jt landed; stores issued a tile ago acknowledged
        __syncthreads();                                         // everyone's pieces landed; the other buffer is free
        if (jt > t0) store_tile(jt - 1);
        if (jt + 1 < t1) dma_tile(lds + ((jt + 1 - t0) & 1) * TILE, bh, (size_t)z * Sp + (size_t)(jt + 1) * KT, tid, wave);
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        // B fragments one k-step ahead of the MFMAs that consume them (LDS latency under the matrix pipe)
        f16x8 bf[2][2];
        const unsigned char* brow[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) brow[ct] = cur + (32 * ct + l31) * ROWB;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) bf[0][ct] = *reinterpret_cast<const f16x8*>(brow[ct] + ((h ^ (l31 & 15)) * 16));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    bf[(s + 1) & 1][ct] = *reinterpret_cast<const f16x8*>(brow[ct] + (((2 * (s + 1) + h) ^ (l31 & 15)) * 16));
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bf[s & 1][ct], acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float cl = cstat[(jt - t0) * KT + 32 * ct + l31];
            bool cmasked = false;
            if (MASKS && mask1) {
                const int j = jt * KT + 32 * ct + l31;
                cmasked = j < S && !mask1[(size_t)z * S + j];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = fmaf(acc[ct][r], c2, -(rl[r] + cl));
                if (MASKS && (((rmask_bits >> r) & 1u) || cmasked)) e = fill2x2 - (rl[r] + cl);
                hold[ct][r] = __builtin_amdgcn_exp2f(e);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_tile(t1 - 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// k1_conf_wide: the same arithmetic with the operand roles swapped so that a workgroup's stores of one step cover
// 32 rows x 1 KiB CONTIGUOUS (k1_conf above: 128 rows x 256 B): each wave keeps the fragments of 64 COLUMNS in registers
// (128 VGPRs) -- the four waves sit side by side, 256 columns -- and 32-row tiles of the other map stream through LDS.
// HBM pages (1-2 KiB of one matrix row) are then written within one step instead of over eight.
// Work item = (pair, 256-column block, chunk of row tiles).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RT = 32;                  // rows per tile
constexpr int RTILE = RT * ROWB;        // 16 KiB
constexpr int RCHUNK_MAX = 32;          // row tiles per work item
constexpr int RING = 4;                 // LDS slots of 32-row tiles (requests run three tiles ahead)

// LDS-DMA written as inline asm ON PURPOSE: hipcc does not see these requests, so it neither counts them nor guards the
// ds_reads of the tile with its own wait.  With the builtin it must assume that a ds_read may alias ANY pending LDS-DMA
// -- including the one just issued for the other buffer -- and emits s_waitcnt vmcnt(0) in front of the first MFMA,
// draining the stores this pipeline wants in flight.  Completion is counted by hand (vmcnt, see the loop), visibility to
// the other waves by the barrier.  M0 (the LDS base of the request) is saved and restored inside the statement
// (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// The stores too: hipcc keeps a store's DATA registers reserved until vmcnt says the store has completed, i.e. it puts
// s_waitcnt vmcnt(31) ... vmcnt(0) in front of the MFMAs / ds_reads that reuse them -- again draining what should stay in
// flight.  The hardware has read a dword store's data by the time the next instruction issues (only 96/128-bit stores
// need wait states, cdna_hip_programming.md section 5.7), so the stores are written as asm and counted by hand.
__device__ __forceinline__ void gstore32(void* addr, float v) {
    asm volatile("global_store_dword %0, %1, off" : : "v"(addr), "v"(v) : "memory");
}
// saddr form: address = wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset + immediate.  No VALU address
// arithmetic per store (three 64-bit vector adds per store were the largest non-matrix cost of the writer's loop).
// NOP5: five wait states in front (an SALU write of the base SGPRs may sit directly before the statement; hipcc pads
// nothing inside asm, and a VMEM instruction reading a just-written SGPR base is a hazard that hung the GPU here).
template <int IMM, bool NOP5 = false>
__device__ __forceinline__ void gstore32_s(const void* sbase_uniform, unsigned voff, float v) {
    if (NOP5)
        asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2 offset:%3 nt" : : "v"(voff), "v"(v), "s"(sbase_uniform), "n"(IMM) : "memory");
    else
        asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" : : "v"(voff), "v"(v), "s"(sbase_uniform), "n"(IMM) : "memory");
}

__device__ __forceinline__ void dma_rows(unsigned char* lds, const _Float16* g, size_t row0, int tid, int wave) {
    const unsigned char* s = reinterpret_cast<const unsigned char*>(g + row0 * C) + tid * 16;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)(lds + wave * 1024));
#pragma unroll
    for (int j = 0; j < RTILE / 4096; ++j) glds16(s + j * 4096, dst + j * 4096);
}

template <bool MASKS>
__global__ __launch_bounds__(256, 2) void k1_conf_wide(const _Float16* __restrict__ ah, const _Float16* __restrict__ bh,
                                                       int Z, int L, int S, int Lp, int Sp, int nJ, int nch, int tiles_per_chunk,
                                                       float c2, float fill2x2, const uint8_t* __restrict__ mask0,
                                                       const uint8_t* __restrict__ mask1, const float2* __restrict__ rowstat,
                                                       const float* __restrict__ cmax, const float* __restrict__ cinv,
                                                       float* __restrict__ conf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    int z, item;
    tile_coords(nJ * nch, Z, z, item);
    const int Jb = item / nch, ch = item - Jb * nch;
    const int ntile = (L + RT - 1) / RT;
    const int t0 = ch * tiles_per_chunk, t1 = min(ntile, t0 + tiles_per_chunk);
    if (t0 >= t1) return;
    const int col0 = Jb * 256 + 64 * wave;                       // first column of this wave (may be >= S: dead wave)
    // B operand: this wave's 64 columns, all 256 channels, in registers (128 VGPRs); lane = (column l31 of subtile ct, k-half h)
    f16x8 bfr[2][NS];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int j = min(col0 + 32 * ct + l31, Sp - 1);
        const _Float16* p = bh + ((size_t)z * Sp + j) * C;
#pragma unroll
        for (int s = 0; s < NS; ++s) bfr[ct][s] = *reinterpret_cast<const f16x8*>(p + 8 * ((2 * s + h) ^ (j & 15)));
    }
    // column terms of this lane's two columns (padded / out-of-range columns: +inf -> p = 0, never stored)
    float cl[2];
    bool cmasked[2] = {false, false};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int j = col0 + 32 * ct + l31;
        cl[ct] = j < S ? cmax[(size_t)z * Sp + j] - __builtin_amdgcn_logf(cinv[(size_t)z * Sp + j]) : 1.0e30f;
        if (MASKS && mask1) cmasked[ct] = j < S && !mask1[(size_t)z * S + j];
    }
    const bool cols_live = col0 < S;                              // wave-uniform
    const bool cols_full = col0 + 64 <= S;                        // wave-uniform
    // this chunk's row terms -> LDS  (rows >= L: 0, never stored)
    float* const rstat = reinterpret_cast<float*>(lds + RING * RTILE);         // [RCHUNK_MAX * RT]
    for (int o = tid; o < (t1 - t0) * RT; o += 256) {
        const int i = t0 * RT + o;
        float v = 0.f;
        if (i < L) { const float2 st = rowstat[(size_t)z * L + i]; v = st.x + __builtin_amdgcn_logf(st.y); }
        rstat[o] = v;
    }
    // wave-uniform base of this wave's first column (readfirstlane: `wave` = tid >> 6 is not provably uniform to hipcc)
    const unsigned long long cb = reinterpret_cast<unsigned long long>(conf + (size_t)z * L * S + col0);
    unsigned char* const cbase = reinterpret_cast<unsigned char*>(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(cb >> 32)) << 32) |
        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cb));     // (the builtin returns a signed int)
    const unsigned lane_off = (unsigned)(4 * h * S + l31) * 4u;
    const size_t row_bytes = (size_t)S * 4;
    unsigned offr[16];                                            // byte offset of (row mfma32_row(r, h), column l31) in a tile
#pragma unroll
    for (int r = 0; r < 16; ++r) offr[r] = (unsigned)(mfma32_row(r, h) * S + l31) * 4u;
    // Everything loaded so far is made "arrived" HERE as far as hipcc's wait bookkeeping goes (an empty asm that uses the
    // registers): otherwise it carries the 32 fragment loads as pending into the loop and re-executes their lazy waits
    // (s_waitcnt vmcnt(31), (15), (14) ... (0) in front of the MFMAs that first use each fragment) in EVERY iteration,
    // where they would drain the stores in flight.
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(bfr[ct][s]));
        asm volatile("" : "+v"(cl[ct]));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(offr[r]));   // keep them in registers (not rematerialised per tile)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the rowstat loads feeding rstat: really arrived)
    // Pipeline per 32-row tile, ring of RING = 4 LDS slots:
    //     wait -> barrier -> request tile it+3 (4 LDS-DMA ops) -> STORE tile it-1 (32 ops) -> MFMAs + exp of tile it.
    // vmcnt retires in issue order.  At the top of iteration `it` the operations issued after the requests of tile `it`
    // are: [stores it-4] req it+1, stores it-3, req it+2, stores it-2  (>= 63 from the fourth iteration on), so
    // s_waitcnt vmcnt(63) -- "all but my 63 youngest operations" -- guarantees that tile `it` has landed while the stores
    // of the last two tiles stay in flight, and the tile requests have three iterations (~10 us) to come back through a
    // fabric saturated with writes (with a one-tile lookahead and vmcnt(0) the same kernel ran at 4.1 TB/s; the store
    // pattern alone sustains 5.5 TB/s, tools/ubench/store_pattern.hip).  Waves whose store bursts are predicated (edge
    // columns / the last, partial row tile) issue an unknown number of stores and simply wait for vmcnt(0).
    const bool exact_counts = cols_full && t1 * RT <= L;            // wave-uniform
#pragma unroll
    for (int k = 0; k < RING - 1; ++k)
        if (t0 + k < t1) dma_rows(lds + k * RTILE, ah, (size_t)z * Lp + (size_t)(t0 + k) * RT, tid, wave);
    // one register set serves as MFMA accumulator AND as the tile kept for the late stores: the stores of tile it-1 are
    // issued before the first MFMA of tile it (which starts from C = 0) overwrites it
    f32x16 acc[2];
    auto store_tile = [&](int it) {
        unsigned char* const tbase = cbase + (size_t)it * RT * row_bytes;
        const int i0 = it * RT;
        if (cols_full && i0 + RT <= L) {
            // ONE scalar base per tile (written many instructions before its first use: an SALU write directly in front of
            // a VMEM instruction that reads the SGPR as its base is a hazard hipcc does not pad inside asm -- a per-row
            // s_add in front of each store hung the GPU), the row in the per-lane offset register offr[r], the column
            // half in the immediate.
            gstore32_s<0, true>(tbase, offr[0], acc[0][0]);
            gstore32_s<128>(tbase, offr[0], acc[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                gstore32_s<0>(tbase, offr[r], acc[0][r]);
                gstore32_s<128>(tbase, offr[r], acc[1][r]);
            }
        } else if (cols_live) {                                     // edge tiles: the same stores, predicated per lane
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool rok = i0 + mfma32_row(r, h) < L;
                if (rok && col0 + l31 < S) gstore32_s<0, true>(tbase, offr[r], acc[0][r]);
                if (rok && col0 + 32 + l31 < S) gstore32_s<128, true>(tbase, offr[r], acc[1][r]);
            }
        }
    };
    for (int it = t0; it < t1; ++it) {
        const int k = it - t0;
        const unsigned char* cur = lds + (k & (RING - 1)) * RTILE;
        // `need` = operations this wave issued after the requests of tile it: 4 per later tile request (tiles k+1 .. k+2 that
        // exist) + 32 per store burst (issued in the iterations between that request and now); any immediate <= need is safe
        int need = 0;
        if (exact_counts) {
            const int n = t1 - t0;
            const int later = min(k + RING - 2, n - 1) - k;
            const int bursts = k >= RING - 1 ? (k - 1) - max(k - (RING - 1), 1) + 1 : max(k - 1, 0);
            need = 4 * later + 32 * bursts;
        }
        if (need >= 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
        else if (need >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (need >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (need >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // everyone's pieces landed; slot (k + 3) % 4 is free
        if (it + RING - 1 < t1)
            dma_rows(lds + ((k + RING - 1) & (RING - 1)) * RTILE, ah, (size_t)z * Lp + (size_t)(it + RING - 1) * RT, tid, wave);
        if (it > t0) store_tile(it - 1);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        // A fragments (row l31 of the tile) one k-step ahead of the MFMAs that consume them
        const unsigned char* arow = cur + l31 * ROWB;
        f16x8 af[2];
        af[0] = *reinterpret_cast<const f16x8*>(arow + ((h ^ (l31 & 15)) * 16));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) af[(s + 1) & 1] = *reinterpret_cast<const f16x8*>(arow + (((2 * (s + 1) + h) ^ (l31 & 15)) * 16));
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1], bfr[ct][s], acc[ct], 0, 0, 0);
        }
        // row terms of this lane's 16 rows: i = it*32 + (r & 3) + 8 (r >> 2) + 4 h  ->  four 16-byte LDS reads
        float rl[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(rstat + (it - t0) * RT + 8 * q + 4 * h);
            rl[4 * q + 0] = a.x; rl[4 * q + 1] = a.y; rl[4 * q + 2] = a.z; rl[4 * q + 3] = a.w;
        }
        unsigned rmask_bits = 0;
        if (MASKS && mask0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = it * RT + mfma32_row(r, h);
                if (i < L && !mask0[(size_t)z * L + i]) rmask_bits |= 1u << r;
            }
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = fmaf(acc[ct][r], c2, -(rl[r] + cl[ct]));
                if (MASKS && (((rmask_bits >> r) & 1u) || cmasked[ct])) e = fill2x2 - (rl[r] + cl[ct]);
                acc[ct][r] = __builtin_amdgcn_exp2f(e);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_tile(t1 - 1);
}

// one thread per slot LIST written by k1_rowstats<CAND> (column j, half-wave h): its entries (i, bits of the
// split-precision log2-domain score x) -> the fused matcher's formula (dual_softmax_f16s.hip:k1_match), bit for bit
__global__ __launch_bounds__(256) void k1_conf_fix(int Z, int L, int S, int Sp, int slots, const float2* __restrict__ rowstat,
                                                   const float* __restrict__ cmax, const float* __restrict__ cinv,
                                                   float* __restrict__ conf, const int* __restrict__ fix_count,
                                                   const uint2* __restrict__ fix_list, int* __restrict__ fix_info_out) {
    const size_t total = (size_t)Z * S * 2;
    int done = 0, dropped = 0;
    for (size_t lh = (size_t)blockIdx.x * blockDim.x + threadIdx.x; lh < total; lh += (size_t)gridDim.x * blockDim.x) {
        const int cnt = fix_count[lh];                    // (z S + j) * 2 + h
        if (cnt <= 0) continue;
        if (cnt > slots) dropped += cnt - slots;
        const size_t zj = lh >> 1, z = zj / (size_t)S;
        const int j = (int)(zj - z * S);
        const float cm = cmax[z * Sp + j], ci = cinv[z * Sp + j];
        for (int k = 0; k < min(cnt, slots); ++k) {
            const uint2 ent = fix_list[lh * slots + k];
            const size_t zi = z * L + ent.x;
            const float2 st = rowstat[zi];
            const float x2 = 2.0f * __uint_as_float(ent.y);
            conf[zi * S + j] = __builtin_amdgcn_exp2f((x2 - st.x) - cm) * (1.0f / st.y) * ci;
            ++done;
        }
    }
    if (fix_info_out) {
        if (done) atomicAdd(&fix_info_out[0], done);
        if (dropped) atomicAdd(&fix_info_out[1], dropped);
    }
}

}  // namespace far_conf

// Called by far_conf_matrix_f16s (dual_softmax_f16s.hip) after the statistics passes.  ah / bh: fp16 `hi` planes of
// k1_prep; c1: log2-domain score per unit of the pre-scaled dot product; fix_count / fix_list: the per-column slot lists
// written by k1_rowstats<CAND>.
int far_k1_conf_launch(const _Float16* ah, const _Float16* bh, int Z, int L, int S, int Lp, int Sp, float c1, float fill2,
                       const uint8_t* mask0, const uint8_t* mask1, const float2* rowstat, const float* cmax,
                       const float* cinv, float* conf, const int* fix_count, const uint2* fix_list, int slots,
                       int* fix_info_out, hipStream_t stream) {
    using namespace far_conf;
    const float c2 = 2.0f * c1, f2 = 2.0f * fill2;
    if (far_get_tuning(2) == 1) {          // A/B experiment: the tall-tile writer (128 rows x 64 columns per step)
        const int nI = Lp / 128;
        const int ntile = (S + KT - 1) / KT;
        int nch = (ntile + 14) / 15;
        const int tpc = (ntile + nch - 1) / nch;                     // <= 15 <= CHUNK_MAX
        nch = (ntile + tpc - 1) / tpc;
        const size_t smem = 2 * TILE + CHUNK_MAX * KT * sizeof(float);
        FAR_ONCE_PER_DEVICE(
            hipFuncSetAttribute((const void*)k1_conf<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipFuncSetAttribute((const void*)k1_conf<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const dim3 grid((unsigned)(Z * nI * nch));
        if (mask0 || mask1)
            hipLaunchKernelGGL(k1_conf<true>, grid, dim3(256), smem, stream, ah, bh, Z, L, S, Lp, Sp, nI, nch, tpc, c2, f2, mask0,
                               mask1, rowstat, cmax, cinv, conf);
        else
            hipLaunchKernelGGL(k1_conf<false>, grid, dim3(256), smem, stream, ah, bh, Z, L, S, Lp, Sp, nI, nch, tpc, c2, f2, mask0,
                               mask1, rowstat, cmax, cinv, conf);
    } else {
        // work item = (pair, 256-column block, chunk of <= 30 row tiles): ~6 rounds of the chip's 512 resident workgroups
        const int nJ = (S + 255) / 256;
        const int ntile = (L + RT - 1) / RT;
        const int want = far_get_tuning(3) > 0 ? far_get_tuning(3) : 30;     // row tiles per item (A/B knob; default 30)
        int nch = (ntile + want - 1) / want;
        const int tpc = (ntile + nch - 1) / nch;                     // <= RCHUNK_MAX
        nch = (ntile + tpc - 1) / tpc;
        const size_t smem = RING * RTILE + RCHUNK_MAX * RT * sizeof(float);
        FAR_ONCE_PER_DEVICE(
            hipFuncSetAttribute((const void*)k1_conf_wide<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipFuncSetAttribute((const void*)k1_conf_wide<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const dim3 grid((unsigned)(Z * nJ * nch));
        if (mask0 || mask1)
            hipLaunchKernelGGL(k1_conf_wide<true>, grid, dim3(256), smem, stream, ah, bh, Z, L, S, Lp, Sp, nJ, nch, tpc, c2, f2,
                               mask0, mask1, rowstat, cmax, cinv, conf);
        else
            hipLaunchKernelGGL(k1_conf_wide<false>, grid, dim3(256), smem, stream, ah, bh, Z, L, S, Lp, Sp, nJ, nch, tpc, c2, f2,
                               mask0, mask1, rowstat, cmax, cinv, conf);
    }
    if (fix_info_out) hipMemsetAsync(fix_info_out, 0, 2 * sizeof(int), stream);
    hipLaunchKernelGGL(k1_conf_fix, dim3(1024), dim3(256), 0, stream, Z, L, S, Sp, slots, rowstat, cmax, cinv, conf, fix_count,
                       fix_list, fix_info_out);
    return far_check_launch();
}
