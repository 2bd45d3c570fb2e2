// EXPERIMENT, not part of libfar_hip.so (round 4).  Measured 0.92x of K17 on every bench shape (128->128 @240x320 x 64 images: 3.25-3.4 ms
// against 3.0-3.1; errors 2-6e-7 of max|ref| vs float64, deterministic): the wiring (C ABI, far_amd/ops.py:PackedWino1d, tools/wino1d_ab.py,
// tools/w1d_timing.py) is in git revision f7158f3; DESIGN.md section 4 (K17, 'what was tried') has the numbers and the reasons.
//
// K18: the stride-1 3x3 convolutions as ONE-DIMENSIONAL Winograd F(2, 3) along x, direct along y, on the f16 matrix cores with
// split-precision operands -- 12 products per 2 outputs where the direct form (K9) has 18 and the 2-D form (K17) has 8, but only
// 4 accumulator planes per 2 outputs (K17: 16 per 4), so a workgroup holds 512 outputs x 64 channels in the registers K17 needs for
// 256 x 64, streams 0.375x the weight bytes per output from L2, requests its operands 4-6 intervals ahead (K17: 1.8) and needs no
// cross-wave exchange in the epilogue.  (K17's measured limit is the operand stream, not the matrix pipe: DESIGN.md section 4.)
//
// Replaces, for inference, what K9 / K17 run for
//   mp3d_loftr/src/loftr/backbone/resnet_fpn.py:5-12   (conv3x3, stride 1)
//                                               :15-43 (BasicBlock: conv -> bn -> relu -> conv -> bn -> +x -> relu)
//                                               :101-119 (layer*_outconv2: conv3x3 -> bn -> leaky_relu -> conv3x3)
//
// Algebra (F(2, 3), cross-correlation form as torch's conv2d).  For the output pair (y, 2t), (y, 2t + 1) of one (ci, co):
//     V_nu[r][t] = (B^T d)_nu  of the four input pixels d_j = x[r][2t - 1 + j]:   V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3
//     U_nu[ky]   = (G g[ky][.])_nu,  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//     M_nu[y][t] = sum_ci sum_ky V_nu[y + ky - 1][t] U_nu[ky]            (per nu: a 3-tap vertical convolution = a GEMM with K = 3 Cin)
//     out[y][2t] = M0 + M1 + M2,   out[y][2t + 1] = M1 - M2 - M3
// U is formed in float64 when the weights are packed, scaled by a power of two and split (hi, lo) into fp16; V is formed in fp32
// from the fp32 activations (one add per element) and split in registers; every product is hi.hi + hi.lo + lo.hi with fp32
// accumulation (fp32-grade), as in K9 / K17.
//
// Tiling (gfx950).  Workgroup = 8 waves = 16 output rows x 32 output columns x 64 output channels, one workgroup per CU.  Wave w owns
// output rows 2w, 2w + 1 and all 16 column pairs: 32 positions (m = 16 rr + t) x 4 nu x 2 channel tiles = 8 accumulator tiles of
// 32x32x16 MFMAs (128 registers) -- all four nu of a position in ONE lane, so the output transform is in-lane.
//  * unit u = (k-step k of 16 input channels, tap row ky): lane (position m, k-group h) reads the four pixels of input row
//    2w + rr + ky - 1 for its eight channels (8 ds_read_b128), forms V0..V3 (32 scalar fp32 adds) and splits them: the result IS the
//    lane's MFMA A operand for the unit's 24 MFMAs (4 nu x 2 channel tiles x 3); the weights U[.][ky] of the unit are one 16 KiB
//    slab [nu][co tile][plane][lane][8] in execution order (the global image is the LDS image).
//  * schedule as K17: waves 0-3 (P) and 4-7 (Q) -- paired on the SIMDs -- alternate every interval: P multiplies unit u while Q
//    transforms unit u, then Q multiplies unit u while P transforms unit u + 1; one raw s_barrier per interval; the LDS-DMA
//    requests ride behind the MFMAs of the multiplying group and are waited for with counted vmcnt (derivation at the K loop).
//  * raw fp32 input: per k-step the 18 x 34 pixel patch as two half patches (rows 0-9 for the P waves, 8-17 for the Q waves; 22.5 KiB
//    each, 2-slot rings), rows padded to 144 16-byte chunks, even / odd columns apart, the channel quad XOR-swizzled by
//    (column pair >> 2) & 3 on the SOURCE address (the LDS image of a request is lane-linear): the transform's ds_read_b128 are
//    conflict free (each 16-lane service group holds 16 distinct column pairs).
//  * weights: 4-slot ring of 16 KiB slabs, requested three units (six intervals) ahead.
// LDS: 4 x 16 KiB + 4 x 23 KiB = 156 KiB.
#include "common.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int SLAB = 16384;                 // weights of one unit: 4 nu x 2 co tiles x 2 planes x 1 KiB
constexpr int NS = 4;                       // weight ring slots
constexpr int ROWCH = 144;                  // 16-byte chunks per patch row (34 pixels x 4 quads = 136, padded: row stride = 0 mod 16 chunks)
constexpr int HALF_ROWS = 10;
constexpr int RAW_CHUNKS = HALF_ROWS * ROWCH;           // 1440
constexpr int RAW_PIECES = (RAW_CHUNKS + 63) / 64;      // 23 wave-DMAs of 1 KiB
constexpr int RAWB = RAW_PIECES * 1024;                 // 23552
constexpr int RAW_OFF = NS * SLAB;
#ifdef FAR_W1D_TIMING
constexpr int SMEM = 163840;
#else
constexpr int SMEM = RAW_OFF + 4 * RAWB;                // 159744 of the CU's 163840
#endif
constexpr int NRP = 6;                                  // raw pieces per wave and half patch (24 >= 23: one repeats)

struct Wino1dArgs {
    const float* x;
    const unsigned char* w;      // packed image (far_wino1d_pack_*): [co block][unit 3 k + ky][16 KiB]
    const float* zeros;          // >= 16 zero bytes (the end of the packed image)
    const float* scale;          // [Cout], includes 2^-(w_exp + 4)
    const float* shift;          // [Cout] or null
    const float* res;            // residual (y's layout) or null
    float* y;
    int* overflow;
    long ntb;                    // tile blocks = N * tilesX * tilesY
    int H, W, Cin, Cout, nk, ncb, tilesX, tilesY, act;
    float slope, out_mul;
};

#ifndef FAR_W1D_EXP
#define FAR_W1D_EXP 0       // experiment builds only: 1 no transform, 2 no MFMAs, 4 no weight requests, 8 no raw requests, 16 no epilogue,
#endif                      // 32 every wait drains the queue (vmcnt(0))

// LDS-DMA from inline asm (invisible to the compiler's wait insertion; completion is counted by hand in the K loop)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// hi = fp16(x), lo = fp16(x - hi) of two values: v_cvt_pk_f16_f32 + one v_fma_mix per lo half (3 instructions per pair)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const f16x2 hh = __builtin_convertvector(f32x2{a, b}, f16x2);
    hi = __builtin_bit_cast(unsigned, hh);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(b));
    lo = l;
}

#ifdef FAR_W1D_TIMING
// Development instrumentation (tools/w1d_timing.py; never defined in the product build): s_memtime stamps of the six units of k-steps
// 2 and 3, kept in the spare 4 KiB of LDS (no memory traffic that would disturb the request queue) and copied out at the end: per wave
// 32 stamps: [4 (u - 6) + e], e = 0 first half done, 2 second half done, 3 barrier passed.
__device__ unsigned long long g_w1d_stamps[4096 * 8 * 32];
#define W1D_T(u, e) do { if ((u) >= 6 && (u) < 12 && (threadIdx.x & 63) == 0) \
    *reinterpret_cast<volatile unsigned long long*>(smem + 159744 + (threadIdx.x >> 6) * 256 + (4 * ((u) - 6) + (e)) * 8) = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W1D_T(u, e) do {} while (0)
#endif
#ifndef W1D_PRIO_MMA
#define W1D_PRIO_MMA 0
#define W1D_PRIO_TR 0
#endif
#define W1D_WAIT(n) do { if (FAR_W1D_EXP & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); } while (0)
#define W1D_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool Q>
__device__ __forceinline__ void w1d_body(const Wino1dArgs& p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Bs = smem;
    unsigned char* const Rs = smem + RAW_OFF + (Q ? 2 * RAWB : 0);        // this group's two half-patch slots

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wsel = wave & 3;                       // this wave among the four of its group; its output rows: 2 wave, 2 wave + 1
    const int l31 = lane & 31, h = lane >> 5;
    const int xt = l31 & 15, rr = l31 >> 4;

    // ---- tile block / channel block: each XCD (block b -> XCD b % 8, speed only) gets a contiguous range of tile blocks and runs the
    // channel blocks of a tile block back to back (they re-read the same pixels from its L2)
    long t;
    int cb;
    {
        const long b = blockIdx.x;
        long seq = b, t0 = 0;
        if ((p.ntb & 7) == 0) { seq = b >> 3; t0 = (b & 7) * (p.ntb >> 3); }
        t = t0 + seq / p.ncb;
        cb = (int)(seq % p.ncb);
    }
    const int bx = (int)(t % p.tilesX);
    t /= p.tilesX;
    const int by = (int)(t % p.tilesY);
    const int img = (int)(t / p.tilesY);
    const int oy0 = 16 * by, ox0 = 32 * bx;
    const int nk = p.nk, nunit = 3 * nk;

    // ---- requests.  Weight slab of unit v: 16 pieces of 1 KiB; the P waves request pieces 0-7 (two per wave) in the even interval
    // of unit v - 3, the Q waves pieces 8-15 in its odd interval, into ring slot v % 4.
    const int wp0 = (Q ? 8 : 0) + 2 * wsel;
    const unsigned char* const wbase = p.w + (size_t)cb * nunit * SLAB + wp0 * 1024 + lane * 16;
    const unsigned bs_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)(Bs + wp0 * 1024));
    const unsigned rs_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)Rs);
    auto w_piece = [&](int unit, int j) {             // piece j (0, 1) of this wave, slab `unit` (past the end: the last slab again)
        const int v = unit < nunit ? unit : nunit - 1;
        if (!(FAR_W1D_EXP & 4)) glds16(wbase + (size_t)v * SLAB + j * 1024, bs_base + (unit & (NS - 1)) * SLAB + j * 1024);
    };

    // ---- raw half patch of a k-step: chunk S = 144 row + 72 (col & 1) + 4 (col >> 1) + (quad ^ (((col >> 1) >> 2) & 3)); piece
    // pc = 4 j + wsel, j < 6 (a wave's slot past the 23 pieces repeats its first piece: harmless, and every wave issues exactly
    // six per half patch, which the counted waits rely on)
    const char* rsrc[NRP];
    int rinc[NRP], rpiece[NRP];
    unsigned rtailm = 0;                            // bit j: the lane's quad lies beyond Cin in the last k-step
    const int rem_ch = p.Cin - 16 * (nk - 1);       // channels of the last k-step (1..16)
#pragma unroll
    for (int j = 0; j < NRP; ++j) {
        int pc = 4 * j + wsel;
        if (pc >= RAW_PIECES) pc = wsel;
        rpiece[j] = pc;
        const int S = pc * 64 + lane;
        const int row = S / ROWCH, rem = S - row * ROWCH;
        const int par = rem >= 72 ? 1 : 0, r2 = rem - 72 * par;
        const int cxi = r2 >> 2, qs = r2 & 3;
        const int quad = qs ^ ((cxi >> 2) & 3);
        const int col = 2 * cxi + par;
        const int iy = oy0 - 1 + row + (Q ? 8 : 0), ix = ox0 - 1 + col;
        const bool ok = S < RAW_CHUNKS && cxi < 17 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const long pix = ((long)img * p.H + iy) * p.W + ix;
        rsrc[j] = ok ? reinterpret_cast<const char*>(p.x + pix * p.Cin + 4 * quad) : reinterpret_cast<const char*>(p.zeros);
        rinc[j] = ok ? 64 : 0;
        if (4 * quad >= rem_ch) rtailm |= 1u << j;
    }
    auto r_piece = [&](int rk, int j) {             // piece j (0..5) of raw half patch rk -> slot rk & 1
        const int kk = rk < nk ? rk : nk - 1;       // past the end: the last patch again, into a slot nobody reads any more
        const bool tail = kk == nk - 1 && ((rtailm >> j) & 1u);
        const char* s = tail ? reinterpret_cast<const char*>(p.zeros) : rsrc[j] + (long)kk * rinc[j];
        if (!(FAR_W1D_EXP & 8)) glds16(s, rs_base + (rk & 1) * RAWB + rpiece[j] * 1024);
    };

    // ---- transform addressing: lane (column pair xt, row rr of the wave's two, k-group h) reads columns 2 xt + j, j = 0..3
    int aoff[4][2];                                 // [column j][16-byte half of the lane's 8 channels]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cx = xt + (j >> 1);
        const int sw = (cx >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) aoff[j][s] = (72 * (j & 1) + 4 * cx + ((2 * h + s) ^ sw)) * 16;
    }
    const int arow = (2 * wsel + rr) * (ROWCH * 16);
    // A operands of the unit's four nu as packed fp16 pairs: element q of Ah[nu] = channels 2q, 2q + 1 of the lane's eight
    u32x4 Ah[4], Al[4];
    auto transform = [&](int rk, int ky) {
        if (FAR_W1D_EXP & 1) {
            if (rk < 0) {
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { Ah[nu][e] = 0x3c003c00u + lane + e; Al[nu][e] = 0x14001400u + nu; }
            }
            return;
        }
        __builtin_amdgcn_s_setprio(W1D_PRIO_TR);
        const unsigned char* R0 = Rs + (rk & 1) * RAWB + arow + ky * (ROWCH * 16);
        f32x4 raw[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            raw[j][0] = *reinterpret_cast<const f32x4*>(R0 + aoff[j][0]);
            raw[j][1] = *reinterpret_cast<const f32x4*>(R0 + aoff[j][1]);
        }
        // scalar fp32 instructions from inline asm on purpose (tools/ubench/valu_cost.hip: packed fp32 adds next to a wave that issues
        // MFMAs cost more than the two scalar ones; left to itself the compiler packs every pair)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ch = 2 * q + e;
                const float d0 = raw[0][ch >> 2][ch & 3], d1 = raw[1][ch >> 2][ch & 3], d2 = raw[2][ch >> 2][ch & 3], d3 = raw[3][ch >> 2][ch & 3];
                asm("v_sub_f32 %0, %1, %2" : "=v"(v[0][e]) : "v"(d0), "v"(d2));
                asm("v_add_f32 %0, %1, %2" : "=v"(v[1][e]) : "v"(d1), "v"(d2));
                asm("v_sub_f32 %0, %1, %2" : "=v"(v[2][e]) : "v"(d2), "v"(d1));
                asm("v_sub_f32 %0, %1, %2" : "=v"(v[3][e]) : "v"(d1), "v"(d3));
            }
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                unsigned hh, ll;
                split_pair(v[nu][0], v[nu][1], hh, ll);
                Ah[nu][q] = hh; Al[nu][q] = ll;
            }
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][ct][r] = 0.f;

    const int b_lane = lane * 16;
    // the 24 MFMAs of a unit with this wave's NP requests spread between them (piece(i): request number i)
    auto mma = [&](int slot, auto np_tag, auto&& piece) {
        constexpr int NP = decltype(np_tag)::value;
        if (FAR_W1D_EXP & 2) {
#pragma unroll
            for (int i = 0; i < NP; ++i) piece(i);
            return;
        }
        __builtin_amdgcn_s_setprio(W1D_PRIO_MMA);
        const unsigned char* B = Bs + slot * SLAB + b_lane;
        f16x8 bh[2][2], bl[2][2];
        auto read_b = [&](int nu) {
            const int q = nu & 1;
            bh[q][0] = *reinterpret_cast<const f16x8*>(B + (nu * 4 + 0) * 1024);
            bl[q][0] = *reinterpret_cast<const f16x8*>(B + (nu * 4 + 1) * 1024);
            bh[q][1] = *reinterpret_cast<const f16x8*>(B + (nu * 4 + 2) * 1024);
            bl[q][1] = *reinterpret_cast<const f16x8*>(B + (nu * 4 + 3) * 1024);
        };
        read_b(0);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const int q = nu & 1;
            if (nu + 1 < 4) read_b(nu + 1);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const int ct = m & 1;
                const f16x8 a = __builtin_bit_cast(f16x8, m < 4 ? Ah[nu] : Al[nu]);
                const f16x8 bb = (m >> 1) == 1 ? bl[q][ct] : bh[q][ct];             // hi.hi, hi.lo, lo.hi
                acc[nu][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, acc[nu][ct], 0, 0, 0);
                const int mi = 6 * nu + m;
                bool any = false;
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if ((i < 2 ? 1 + 2 * i : 5 + 3 * (i - 2)) == mi) { piece(i); any = true; }
                if (any) __builtin_amdgcn_sched_barrier(0);        // the request stays behind this MFMA
            }
        }
    };

    // ---- prologue: slabs 0-2, raw half patch 0 (P: and 1; Q requests its half patch 1 in unit 0); the P waves transform unit 0
#pragma unroll
    for (int v = 0; v < 3; ++v) { w_piece(v, 0); w_piece(v, 1); }
#pragma unroll
    for (int j = 0; j < NRP; ++j) r_piece(0, j);
    if (!Q) {
#pragma unroll
        for (int j = 0; j < NRP; ++j) r_piece(1, j);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (FAR_W1D_EXP & 1) transform(-1, 0);
    if (!Q) transform(0, 0);

    // ---- K loop: ONE barrier per unit u = 3 k + ky.  Inside a unit the P waves multiply unit u and then transform unit u + 1, the Q
    // waves transform unit u and then multiply it: the two waves of a SIMD are in opposite phases without a barrier between the halves
    // (a wave's A operands are its own registers).  Requests of a wave in its multiplying half, in this order: its two pieces of slab
    // u + 3 (slot (u + 3) % 4 = the slot slab u - 1 left: its last reader, Q, finished before the barrier that ended unit u - 1);
    // then the six pieces of a raw half patch -- P at ky = 2: patch k + 2 -> slot k % 2 (the P waves read patch k last in unit
    // (k, 1)); Q at ky = 0: patch k + 1 -> slot (k + 1) % 2 (the Q waves read patch k - 1 last in the first half of unit (k - 1, 2)).
    // Any three consecutive units of a wave carry 6 + 6 requests.  Waits at the end of a unit (vmcnt retires in order):
    //   vmcnt(10): slab u + 1, requested three units ago, has landed (younger: the raw pieces behind it in its unit and the requests of
    //     the two units since: 4 + 6);
    //   P at ky = 1, Q at ky = 2: vmcnt(4) -- the half patch of k-step k + 1 has landed before the group's transform of unit (k + 1, 0)
    //     in the next unit (P: second half of unit (k, 2); Q: first half of unit (k + 1, 0)); younger: 2 + 2 slab pieces.
    auto unit = [&](int k, auto ky_tag) {
        constexpr int ky = decltype(ky_tag)::value;
        const int u = 3 * k + ky;
        constexpr int NP = (Q ? ky == 0 : ky == 2) ? 8 : 2;
        auto piece = [&](int i) {
            if (i < 2) w_piece(u + 3, i);
            else r_piece(Q ? k + 1 : k + 2, i - 2);
        };
        if (!Q) {
            mma(u & (NS - 1), std::integral_constant<int, NP>{}, piece);
            __builtin_amdgcn_sched_barrier(0);
            W1D_T(u, 0);
            if (u + 1 < nunit) transform(ky == 2 ? k + 1 : k, ky == 2 ? 0 : ky + 1);
        } else {
            transform(k, ky);
            __builtin_amdgcn_sched_barrier(0);
            W1D_T(u, 0);
            mma(u & (NS - 1), std::integral_constant<int, NP>{}, piece);
        }
        __builtin_amdgcn_sched_barrier(0);
        W1D_T(u, 2);
        if (Q ? ky == 2 : ky == 1) W1D_WAIT(4);
        else W1D_WAIT(10);
        W1D_BARRIER();
        W1D_T(u, 3);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int k = 0; k < nk; ++k) {
        unit(k, std::integral_constant<int, 0>{});
        unit(k, std::integral_constant<int, 1>{});
        unit(k, std::integral_constant<int, 2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing requests landed before the workgroup ends
#ifdef FAR_W1D_TIMING
    if (blockIdx.x < 4096 && (threadIdx.x & 63) < 32)
        g_w1d_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 + (threadIdx.x & 63)] =
            *reinterpret_cast<volatile unsigned long long*>(smem + 159744 + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63) * 8);
#endif
    if (FAR_W1D_EXP & 16) {
        float tsum = 0.f;
        for (int nu = 0; nu < 4; ++nu) for (int ct = 0; ct < 2; ++ct) for (int r = 0; r < 16; ++r) tsum += acc[nu][ct][r];
        if (tsum == 123.456f) p.y[0] = tsum;
        return;
    }

    // ---- activation-range guard (as K9): a finite accumulator set cannot overflow its own sum
    if (p.overflow) {
        float chk = 0.f;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) chk += acc[nu][ct][r];
        const bool bad = !(fabsf(chk) <= FLT_MAX);
        if (__any(bad) && lane == 0) atomicOr(p.overflow, 1);
    }

    // ---- epilogue, in-lane: a lane holds output channel 64 cb + 32 ct + l31 of the positions m = 8 (r >> 2) + 4 h + (r & 3) =
    // (row rr = m >> 4, column pair m & 15): out[2t] = M0 + M1 + M2, out[2t + 1] = M1 - M2 - M3, scale / shift / residual /
    // activation, 4-byte stores -- 128 contiguous bytes per pixel and half-wave.
    const float as = p.act == 2 ? p.slope : 0.f, ab = p.act == 0 ? -__builtin_inff() : 0.f;     // act(v) = max(v, v as + ab)
    const float* __restrict__ resp = p.res;
    float* __restrict__ yp = p.y;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int co = cb * 64 + 32 * ct + l31;
        const bool cok = co < p.Cout;
        const float sc = (cok ? p.scale[co] : 0.f) * p.out_mul;
        const float sh = (cok && p.shift) ? p.shift[co] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                // accumulator registers 4 g .. 4 g + 3: positions 8 g + 4 h + 0..3
            float rv[4][2];
            long off[4];
            bool ok[4][2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 8 * g + 4 * h + e;
                const int oy = oy0 + 2 * wave + (m >> 4), ox = ox0 + 2 * (m & 15);
                off[e] = (((long)img * p.H + oy) * p.W + ox) * p.Cout + co;
                ok[e][0] = cok && oy < p.H && ox < p.W;
                ok[e][1] = cok && oy < p.H && ox + 1 < p.W;
                rv[e][0] = (resp && ok[e][0]) ? resp[off[e]] : 0.f;
                rv[e][1] = (resp && ok[e][1]) ? resp[off[e] + p.Cout] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                float y0 = (acc[0][ct][r] + acc[1][ct][r]) + acc[2][ct][r];
                float y1 = (acc[1][ct][r] - acc[2][ct][r]) - acc[3][ct][r];
                y0 = y0 * sc + sh + rv[e][0];
                y1 = y1 * sc + sh + rv[e][1];
                y0 = fmaxf(y0, __builtin_fmaf(y0, as, ab));
                y1 = fmaxf(y1, __builtin_fmaf(y1, as, ab));
                if (ok[e][0]) yp[off[e]] = y0;
                if (ok[e][1]) yp[off[e] + p.Cout] = y1;
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void k_wino1d(const Wino1dArgs p) {
    if (threadIdx.x >= 256) w1d_body<true>(p);
    else w1d_body<false>(p);
}

// Packs w (read through element strides like K9's pack: s_co, s_ci per channel, s_tap per tap in execution order ky * 3 + kx) into the
// image [co block][unit 3 k + ky][nu][co tile][plane][lane][8] fp16: lane = (co = 64 cb + 32 ct + (lane & 31), k-group lane >> 5),
// element e = input channel 16 k + 8 (lane >> 5) + e; value = (G g[ky][.])_nu * wmul in float64, hi = fp16(value), lo = fp16(value - hi).
__global__ __launch_bounds__(256) void k_wino1d_pack(const float* __restrict__ w, long s_co, long s_ci, long s_tap, int Cin, int Cout,
                                                     int nk, int ncb, const float* __restrict__ wmul_dev, _Float16* __restrict__ out,
                                                     const float* __restrict__ base_scale, float* __restrict__ scale_vec) {
    const float wmul = wmul_dev[0];
    if (scale_vec && blockIdx.x == 0)
        for (int co = threadIdx.x; co < Cout; co += blockDim.x) scale_vec[co] = (base_scale ? base_scale[co] : 1.0f) * wmul_dev[1];
    const long total = (long)ncb * nk * 3 * 16 * 64;           // 16-byte items
    if (blockIdx.x == 0 && threadIdx.x < 16) out[(size_t)total * 8 + threadIdx.x] = (_Float16)0.f;      // the zero row
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int lane = (int)(t & 63); t >>= 6;
        const int plane = (int)(t & 1); t >>= 1;
        const int ct = (int)(t & 1); t >>= 1;
        const int nu = (int)(t & 3); t >>= 2;
        const int ky = (int)(t % 3); t /= 3;
        const int k = (int)(t % nk);
        const int cb = (int)(t / nk);
        const int co = cb * 64 + ct * 32 + (lane & 31);
        _Float16* dst = out + (size_t)i * 8;
        for (int e = 0; e < 8; ++e) {
            const int ci = 16 * k + 8 * (lane >> 5) + e;
            double u = 0.0;
            if (co < Cout && ci < Cin) {
                const float* g = w + (long)co * s_co + (long)ci * s_ci;
                for (int b = 0; b < 3; ++b) u += (double)g[(long)(3 * ky + b) * s_tap] * G[nu][b];
                u *= (double)wmul;
            }
            const _Float16 hh = (_Float16)u;
            dst[e] = plane == 0 ? hh : (_Float16)(u - (double)hh);
        }
    }
}

}  // namespace

extern "C" {

struct far_conv_desc {          // mirrors include/far_hip.h
    const float* x;
    const float* x2;
    const void* packed;
    const float* scale;
    const float* shift;
    const float* res;
    const float* ln_gamma;
    const float* ln_beta;
    const float* post_res;
    const float* up;
    float* y;
    long N;
    int H, W, Cin, Cin1, Cout, ksize, stride;
    int act, split, out_planes, res_group;
    float slope, ln_eps;
    int act_exp;
    int* overflow;
    const float* act_scale_dev;
};

// Bytes of the F(2, 3) image of a [Cout][Cin][3][3] weight (hi + lo planes, + the zero row padding lanes read).
size_t far_wino1d_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    const size_t nk = (Cin + 15) / 16, ncb = (Cout + 63) / 64;
    return ncb * nk * 3 * SLAB + 32;
}

// w: the element of tap 0 of a [Cout][Cin][3][3] weight read through element strides (a contiguous torch weight: 9 Cin, 9, 1);
// scale_in = { 2^w_exp, 2^-(w_exp + 4) } on the device (far_weight_scale_f32: 2^13 <= max|w| 2^w_exp < 2^14, so |G g| 2^w_exp <=
// 1.5 * 2^14 stays in fp16); scale_vec_out[co] = base_scale[co] (1 when NULL) * scale_in[1] (may be NULL).
int far_wino1d_pack_view_scaled_f32(const float* w, long s_co, long s_ci, long s_tap, int Cin, int Cout, const float* scale_in,
                                    void* packed, const float* base_scale, float* scale_vec_out, hipStream_t stream) {
    far_clear_errors();
    if (!w || !packed || !scale_in || far_wino1d_packed_bytes(Cin, Cout) == 0) return FAR_EINVAL;
    const int nk = (Cin + 15) / 16, ncb = (Cout + 63) / 64;
    const long items = (long)ncb * nk * 3 * 16 * 64;
    long blocks = (items + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(k_wino1d_pack, dim3((unsigned)blocks), dim3(256), 0, stream, w, s_co, s_ci, s_tap, Cin, Cout, nk, ncb, scale_in,
                       (_Float16*)packed, base_scale, scale_vec_out);
    return far_check_launch();
}

// y = act(scale[co] * conv3x3(x, W)[co] + shift[co] + res), stride 1, zero padding 1, NHWC fp32 -- far_conv_nhwc_f32's contract for
// ksize = 3, stride = 1, split = 1 with `packed` a far_wino1d_pack_* image; fields this kernel does not serve (x2, ln_*, post_res, up,
// out_planes > 1, res_group > 1, act_scale_dev, act_exp < 0) are rejected with FAR_EINVAL.  Cin % 4 == 0.  The activations are split
// unscaled: |a| <= 32752 (the transformed operand is at most 2 |a|); `scale` is corrected by 2^4 for the 2^-4 it folds.
int far_conv3x3_wino1d_f32(const far_conv_desc* desc, hipStream_t stream) {
    if (!desc) return FAR_EINVAL;
    const far_conv_desc& d = *desc;
    far_clear_errors();
    if (d.N == 0) return FAR_OK;
    if (!d.x || !d.packed || !d.scale || !d.y || d.N < 0 || d.H <= 0 || d.W <= 0 || d.Cin <= 0 || (d.Cin & 3) || d.Cout <= 0 ||
        d.ksize != 3 || d.stride != 1 || !d.split || d.act < 0 || d.act > 2 || (d.act == 2 && !(d.slope >= 0.f && d.slope <= 1.f)) ||
        d.x == d.y || d.x2 || d.Cin1 != d.Cin || d.ln_gamma || d.ln_beta || d.post_res || d.up || d.out_planes != 1 || d.res_group != 1 ||
        d.act_scale_dev || d.act_exp < 0 || d.act_exp > 8)
        return FAR_EINVAL;
    Wino1dArgs a;
    a.x = d.x; a.w = (const unsigned char*)d.packed;
    a.zeros = reinterpret_cast<const float*>((const unsigned char*)d.packed + far_wino1d_packed_bytes(d.Cin, d.Cout) - 32);
    a.scale = d.scale; a.shift = d.shift; a.res = d.res; a.y = d.y; a.overflow = d.overflow;
    a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Cout = d.Cout; a.nk = (d.Cin + 15) / 16; a.ncb = (d.Cout + 63) / 64;
    a.tilesX = (d.W + 31) / 32; a.tilesY = (d.H + 15) / 16;
    a.ntb = d.N * a.tilesX * a.tilesY;
    a.act = d.act; a.slope = d.slope; a.out_mul = 16.0f;
    const long nblk = a.ntb * a.ncb;
    if (nblk > 0x7fffffffL || (long)d.H * d.W * d.Cout > 0x7fffffffL) return FAR_EINVAL;
    bool cfg_failed = false;
    FAR_ONCE_PER_DEVICE(cfg_failed = hipFuncSetAttribute((const void*)k_wino1d, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess);
    if (cfg_failed) return far_check_launch();
    hipLaunchKernelGGL(k_wino1d, dim3((unsigned)nblk), dim3(512), SMEM, stream, a);
    return far_check_launch();
}

#ifdef FAR_W1D_TIMING
int far_w1d_timing_dump(void* host, int nblocks) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w1d_stamps), (size_t)nblocks * 8 * 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -5;
}
#endif

}  // extern "C"
