// K17: Winograd F(2x2, 3x3) convolution on the f16 matrix cores with SPLIT-PRECISION operands -- the stride-1 3x3 layers of the
// ResNet-FPN backbone with 2.25x fewer matrix instructions than K9's direct implicit GEMM (16 products per 2x2 output tile
// instead of 36), same fp32 tensors in and out, same fp32-grade products (hi.hi + hi.lo + lo.hi, fp32 accumulation), same
// fused epilogue (BatchNorm scale / shift, residual, activation).
//
// Replaces, for inference, what K9 (conv_igemm_f16s.hip) runs for
//   mp3d_loftr/src/loftr/backbone/resnet_fpn.py:5-12   (conv3x3, stride 1)
//                                               :15-43 (BasicBlock: conv -> bn -> relu -> conv -> bn -> +x -> relu)
//                                               :101-119 (layer*_outconv2: conv3x3 -> bn -> leaky_relu -> conv3x3)
//
// Algebra (Lavin & Gray, cross-correlation form as torch's conv2d).  For a 2x2 output tile with its 4x4 input patch d
// (rows oy-1 .. oy+2) and the 3x3 filter g of one (ci, co):
//     Y = A^T [ (G g G^T) o (B^T d B) ] A ,   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
//                                             A^T = [1 1 1 0; 0 1 -1 -1]
// so per position p = (xi, nu) of the 4x4 transform domain the channel contraction is a GEMM  M_p[tile][co] = sum_ci V_p[tile][ci] U_p[ci][co].
// U = G g G^T is formed in float64 when the weights are packed, scaled by a power of two and split (hi, lo) into fp16;
// V = B^T d B is formed in fp32 from the fp32 activations (two adds per element) and split in registers.
//
// Tiling (gfx950).  Workgroup = 8 waves = 16x16 output pixels (8x8 Winograd tiles) x 64 output channels, one workgroup per CU.
// Wave (xi, tb) owns transform row xi (its four positions nu = 0..3) of tile block tb (32 tiles = 4 tile rows) for the 64 channels:
// 4 x 2 accumulator tiles of 32x32x16 MFMAs (128 registers).
//  * the row transform of a wave needs only TWO input rows (xi = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3) and yields all
//    four nu of its row: no transform work is duplicated between the waves.  Lane (tile m, k-group h) transforms the eight channels
//    8h .. 8h+7 of its tile, so the split result IS the MFMA A operand of that lane: V never touches LDS.  (xi = 2 computes
//    d1 - d2 = -V; its weights are packed negated.)
//  * raw fp32 input: per 16-channel k-step the 18x18 pixel patch (20.25 KiB) comes in by LDS-DMA (global_load_lds_dwordx4) into
//    a 3-slot ring; even and odd columns are stored apart and the four 16-byte channel quads of a pixel are XOR-swizzled by
//    (row >> 1) & 3 -- on the SOURCE address, the LDS image being lane-linear -- so that the transform's ds_read_b128 (lanes =
//    tiles two pixels apart) are conflict free (measured: SQ_LDS_BANK_CONFLICT = 0; tools/wino_banks.py searches the layouts).
//  * weights: per k-step and transform-row pair {0,1} / {2,3} one 32 KiB half-slab [xi][nu][co tile][plane][lane][8] in execution
//    order (the global image IS the LDS image) into a 3-slot ring two intervals ahead; shared by the two tile blocks.
//  * schedule: the waves 0-3 (xi = 0, 1) and 4-7 (xi = 2, 3) -- paired on the four SIMDs -- alternate roles every interval:
//    one group issues its 24 MFMAs of k-step k, with the LDS-DMA requests behind them, while the other transforms its A operands
//    (16 LDS reads, 64 scalar fp32 adds, 48 conversion instructions); one raw s_barrier per interval, counted vmcnt (the barrier
//    never drains the request queue; the requests come from inline asm so that the compiler's wait insertion does not either).
//  * epilogue: in-lane output transform over nu, the xi sums through LDS (128 KiB, the K-loop buffers are dead), scale / shift /
//    residual / activation, 16-byte stores (256 contiguous bytes per pixel and workgroup).
// Numerics: the input transform adds two roundings of 2^-24 to every A operand and the output transform sums nine products;
// measured against a float64 convolution in tests/test_conv_gpu.py next to K9: 3.5e-7 of max |ref| (K9: 1.2e-6 -- fewer
// accumulation steps per output), bar 2e-6.
// No input scaling: |a| <= 16376 survives the split (|V| <= 4 |a|); beyond it the accumulators turn non-finite and the launch
// raises the activation-overflow flag like K9 (the host then takes K9 with a lower activation exponent).
//
// Round 5 (profiles/r05_k17_ab.txt, docs/rounds/r05.md): four variants were built on this kernel and measured bit-identical but no
// faster -- the channel blocks of a tile block walked inside one workgroup (+1.5 %), non-temporal requests (+3 %), the raw-patch
// requests spread over all eight waves (+0.4 %), the k order rotated per channel block so that sibling workgroups prefetch for
// each other (+1 %) -- and removed again; the issue priority of the multiplying group stayed (-2 ... -3 %).
// STATUS (round 4): dispatched by ops.conv_nhwc for the inference step's stride-1 3x3 layers (ops.USE_WINO): same-box step
// 99.8 -> 94.5 ms.  Against K9 at 64 images (profiles/r04_k17_winograd.txt): 128 -> 128 @240x320 3.04 vs 3.39 ms (1.11x),
// 196 -> 196 @240x320 8.83 vs 9.45 (1.07x), 196 -> 128 1.16x, 256 -> 256 @120x160 1.19x, 256 -> 256 @60x80 1.26x -- short of the
// 2.25x fewer matrix instructions, and of the 1.3x the round set as its bar.  Where the time goes (tools/wino_timing*.py,
// tools/wino_pmc.sh): the matrix pipe is busy 27 %; the K loop WITHOUT its MFMAs still takes 1.85 of the
// 2.5 ms, because a workgroup moves 84 KiB per k-step from L2 into LDS (64 KiB of them transformed weights).  That is the
// structural cost of F(2x2, 3x3): 16 accumulator planes per 4 outputs, so only 256 outputs x 64 channels of accumulators fit in
// half the register file of a CU (K9 holds 1024 x 64 ... per the same registers), and the 16 weight planes are re-streamed for
// every such tile: 26 GB through L2 per launch, ~14 TB/s, 3.5e8 L2 requests in 3 ms.  Every request also costs its wave ~55
// cycles of issue stall, and with 160 KiB of LDS the rings reach one to two intervals ahead of a ~2 us request latency under load.
#include "common.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int SLAB = 32768;                 // weight half-slab: 2 xi x 4 nu x 2 co tiles x 2 planes x 1 KiB
constexpr int RAW_SLOTS = 18 * 18 * 4;      // 16-byte slots of one raw patch (pixel x channel quad)
constexpr int RAW_PIECES = (RAW_SLOTS + 63) / 64;       // 21 wave-DMAs of 1 KiB (the tail of the last one receives zeros)
constexpr int RAWB = RAW_PIECES * 1024;     // raw ring slot
constexpr int RAW_OFF = 3 * SLAB;
#ifdef FAR_WINO_TIMING2
constexpr int SMEM = 163840;
#else
constexpr int SMEM = RAW_OFF + 3 * RAWB;    // 162816 B of the CU's 163840
#endif
constexpr int SMEM_EPI = 8 * 16384;         // Z exchange of the epilogue
static_assert(SMEM_EPI <= SMEM, "epilogue exchange fits the loop buffers");

struct WinoArgs {
    const float* x;
    const unsigned char* w;      // packed image (far_wino_pack_*): [co block][k-step][half][32 KiB]
    const float* zeros;          // >= 16 zero bytes (the end of the packed image)
    const float* scale;          // [Cout], includes 2^-(w_exp + 4)
    const float* shift;          // [Cout] or null
    const float* res;            // residual (y's layout) or null
    float* y;
    int* overflow;
    long ntb;                    // tile blocks = N * tilesX * tilesY
    int H, W, Cin, Cout, nk, ncb, tilesX, tilesY, act;
    int half_ok;                 // the last channel block holds at most 32 channels: its workgroups run the HALF body
    int prio;                    // 1 (default; far_set_tuning(15, 1) clears it): the multiplying wave group raises its issue priority
    float slope, out_mul;
};

// LDS-DMA from inline asm: invisible to the compiler's wait insertion (with the builtin it drains the whole request queue,
// vmcnt(0), in front of the first raw-patch ds_read of every transform); completion is counted by hand in the K loop.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// hi = fp16(x), lo = fp16(x - hi) of two values: v_cvt_pk_f16_f32 + one v_fma_mix per lo half (the fp16 operand is widened,
// subtracted in fp32 -- exactly -- and the result rounded to fp16 by the same instruction): 3 instructions per pair.
template <bool MIX>
__device__ __forceinline__ void split_pair(float a, float b, f16x2& hi, f16x2& lo) {
    if (MIX) {
        hi = __builtin_convertvector(f32x2{a, b}, f16x2);
        const unsigned h = __builtin_bit_cast(unsigned, hi);
        unsigned l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
        // (no wait state behind the half-register write, unlike common.h's split2: every consumer of `lo` in this kernel is an LDS store,
        // which reads the register file, not the VALU's forwarding path the hazard lives in)
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
        lo = __builtin_bit_cast(f16x2, l);
    } else {
        split2(f32x2{a, b}, hi, lo);
    }
}

template <bool MIX>
__device__ __forceinline__ void split8w(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        f16x2 h, l;
        split_pair<MIX>(v[i], v[i + 1], h, l);
        hi[i] = h.x; hi[i + 1] = h.y;
        lo[i] = l.x; lo[i + 1] = l.y;
    }
}

#ifdef FAR_WINO_TIMING
// Development instrumentation (tools/wino_timing.py; never defined in the product build): s_memtime stamps of waves 0 and 4 of
// the first 4096 workgroups (every wave): [0] entry, [1] prologue done, [2 + 2 i] interval i work issued, [3 + 2 i] interval i barrier passed,
// [60] K loop drained, [61] Z exchanged, [62] stores issued, [63] stores acknowledged.
__device__ unsigned long long g_wino_stamps[4096 * 8 * 64];
#define FAR_WINO_STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_wino_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FAR_WINO_STAMP(i) do {} while (0)
#endif
#ifdef FAR_WINO_TIMING2
// Second instrumentation (tools/wino_timing2.py): s_memtime stamps of k-steps 2 and 3 kept in the last KiB of LDS (no memory traffic
// that would disturb the request queue), copied out at the end: per wave 16 stamps: [8 (k - 2) + e], e = 0 interval start, 1 work done,
// 2 requests waited for, 3 barrier passed, 4 work done (odd interval), 5 requests waited for, 6 barrier passed.
__device__ unsigned long long g_wino_stamps2[4096 * 8 * 16];
#define FAR_WINO_T2(k, e) do { if ((k) >= 2 && (k) < 4 && (threadIdx.x & 63) == 0) \
    *reinterpret_cast<volatile unsigned long long*>(smem + 162816 + (threadIdx.x >> 6) * 128 + (8 * ((k) - 2) + (e)) * 8) = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FAR_WINO_T2(k, e) do {} while (0)
#endif
#ifndef FAR_WINO_EXP
#define FAR_WINO_EXP 0      // experiment builds only (tools/wino_exp.sh): 1 no transform, 2 no MFMAs, 4 no weight requests, 8 no raw
#endif                      // requests, 16 no epilogue, 32 every wait drains the queue (vmcnt(0)), 64 raw requests to a cache-resident region, 128 no wait for the prologue's requests

// HALF: the last channel block of a layer whose channel count leaves it at most 32 channels (196 / 208 outputs: 4 or 16 of 64) --
// only the first 32-channel tile is multiplied (12 MFMAs per interval), and the waves whose weight pieces belong to the second
// tile (wave & 2: piece 4 j + wave = [xi & 1][nu][co tile][plane]) request none.
template <bool Q, bool MIX, bool HALF>
__device__ __forceinline__ void wino_body(const WinoArgs& p) {
    constexpr int NCT = HALF ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Bs = smem;
    unsigned char* const Rs = smem + RAW_OFF;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave >> 1, tb = wave & 1, xil = xi & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // ---- tile block / channel block of this workgroup: each XCD (block b -> XCD b % 8, speed only) gets a contiguous range of
    // tile blocks and runs the channel blocks of a tile block back to back (they re-read the same pixels from its L2)
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
    const int oy0 = 16 * by, ox0 = 16 * bx;
    const int nk = p.nk;

    // ---- the epilogue's per-channel vectors (this lane's four channels there), requested now: they have long arrived by then
    f32x4 sc4, sh4;                                 // (x out_mul in the epilogue: no use, and so no wait, here)
    {
        const int c4 = cb * 64 + 4 * (lane & 15);
        sc4 = *reinterpret_cast<const f32x4*>(c4 < p.Cout ? p.scale + c4 : p.zeros);          // unconditional loads: no branch, no wait
        sh4 = *reinterpret_cast<const f32x4*>((c4 < p.Cout && p.shift) ? p.shift + c4 : p.zeros);
    }

    // ---- requests.  All LDS-DMA pieces (1 KiB per wave instruction) are issued by the group that is MULTIPLYING in the
    // interval, behind its MFMAs (the matrix pipe leaves its wave's issue slots free; in front of a transform they cost that
    // wave ~100 cycles apiece): per interval the 32 pieces of one weight half-slab (8 per wave, behind the first eight MFMAs),
    // in the even intervals also the 21 pieces of a raw patch (6 per wave of the xi = 0, 1 group -- the older waves, which win the
    // issue arbitration on their SIMDs and finish their 24 MFMAs first).
    // Weight half-slabs: slab i = 2 k + half is read in interval i, from ring slot i % 3.
    const int wsel = wave & 3;                      // this wave among the four of its group
    const int nslab = 2 * nk;
    const unsigned char* const wbase = p.w + (size_t)cb * nslab * SLAB + wsel * 1024 + lane * 16;
    const unsigned bs_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)(Bs + wsel * 1024));
    const unsigned rs_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)Rs);
    auto slab_src = [&](int slab) {                 // this lane's source of piece 0 of a slab (past the end: the last slab again)
        const int i = slab < nslab ? slab : nslab - 1;
        return wbase + (size_t)i * SLAB;
    };
    const bool wskip = HALF && (wsel & 2);            // wave-uniform
    auto b_piece = [&](const unsigned char* src0, int slot, int j) {          // piece j (0..7) of this wave
        if (wskip) return;
        if (!(FAR_WINO_EXP & 4)) glds16(src0 + j * 4096, bs_base + slot * SLAB + j * 4096);
    };

    // ---- raw patch of a k-step (16 channels of the 18 x 18 pixels) in a 3-slot ring: 16-byte slot S = 4 index' + quad',
    // index' = 18 row + 9 (col & 1) + (col >> 1) (even and odd columns apart), quad' = quad ^ ((row >> 1) & 3); the swizzle is on
    // the SOURCE address (the LDS image of a request is lane-linear).  Piece pc = 4 j + wsel, j < 6, of the xi = 0, 1 waves; a
    // wave's slots past the 21 pieces repeat its first piece (the same request twice: harmless, and every wave issues exactly six,
    // which the counted waits rely on).
    constexpr int NRP = 6;
    const char* rsrc[NRP];
    int rinc[NRP], rpiece[NRP];
    unsigned rtailm = 0;                            // bit j: the lane's quad lies beyond Cin in the last k-step
    const int rem_ch = p.Cin - 16 * (nk - 1);       // channels of the last k-step (1..16)
    if (!Q) {
#pragma unroll
        for (int j = 0; j < NRP; ++j) {
            int pc = 4 * j + wsel;
            if (pc >= RAW_PIECES) pc = wsel;
            rpiece[j] = pc;
            const int S = pc * 64 + lane;
            const int idx = S >> 2, sp = S & 3;
            const int prow = idx / 18, rem = idx - prow * 18;
            const int pcol = rem < 9 ? 2 * rem : 2 * (rem - 9) + 1;
            const int quad = sp ^ ((prow >> 1) & 3);
            const int iy = oy0 - 1 + prow, ix = ox0 - 1 + pcol;
            const bool ok = S < RAW_SLOTS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const long pix = ((long)img * p.H + iy) * p.W + ix;
            rsrc[j] = ok ? reinterpret_cast<const char*>(p.x + pix * p.Cin + 4 * quad) : reinterpret_cast<const char*>(p.zeros);
            rinc[j] = ok ? 64 : 0;
            if (4 * quad >= rem_ch) rtailm |= 1u << j;
        }
    }
    auto raw_piece = [&](int rk, int slot, int j) {                           // piece j (0..5) of raw patch rk
        if (Q) return;
        const int kk = rk < nk ? rk : nk - 1;       // past the end: the last patch again, into a slot nobody reads
        const bool tail = kk == nk - 1 && ((rtailm >> j) & 1u);
        const char* s = tail ? reinterpret_cast<const char*>(p.zeros) : rsrc[j] + (long)kk * rinc[j];
        if (FAR_WINO_EXP & 64) {         // experiment: every workgroup reads the same 128 KiB (cache hits), same request count and shape
            const char* f = reinterpret_cast<const char*>(p.x) + ((rpiece[j] * 1024 + (kk & 3) * 32768 + (threadIdx.x & 63) * 16) & 131071);
            glds16(f, rs_base + slot * RAWB + rpiece[j] * 1024);
        } else
        if (!(FAR_WINO_EXP & 8)) glds16(s, rs_base + slot * RAWB + rpiece[j] * 1024);
    };

    // ---- transform addressing: lane (tile m = l31: row tyl = m >> 3 of the block's four, column tx = m & 7; k-group h)
    const int ra = xi == 0 ? 0 : 1, rb = xi == 3 ? 3 : 2;
    const float sb = xi == 1 ? 1.0f : -1.0f;
    const int ty = 4 * tb + (l31 >> 3), tx = l31 & 7;
    int aoff[2][2];                                 // [row a / b][16-byte half of the lane's 8 channels]
    {
        const int rr[2] = {ra, rb};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int prow = 2 * ty + rr[q];
            const int sg = (prow >> 1) & 3;
            const int base = (prow * 18 + tx) * 64;
            aoff[q][0] = base + (((2 * h) ^ sg) * 16);
            aoff[q][1] = base + (((2 * h + 1) ^ sg) * 16);
        }
    }
    // A operands of the wave's four positions as packed fp16 pairs: element p of Ah[nu] = channels 2p, 2p + 1 of the lane's eight
    u32x4 Ah[4], Al[4];
    auto transform = [&](int slot) {
        if (FAR_WINO_EXP & 1) {
            if (slot < 0) {
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { Ah[nu][e] = 0x3c003c00u + lane + e; Al[nu][e] = 0x14001400u + nu; }
            }
            return;
        }
        const unsigned char* R0 = Rs + slot * RAWB;
        // Scalar fp32 instructions from inline asm on purpose: next to a wave that issues MFMAs, v_pk_fma_f32 waits for a gap in the
        // matrix pipe (tools/ubench/valu_cost.hip: 370 cycles per instruction against 5 alone) and the packed adds cost more than
        // the two scalar ones; left to itself the compiler packs every pair.  All sixteen reads first.
        f32x4 raw[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int co = (c & 1) * (9 * 64) + (c >> 1) * 64;
            raw[c][0] = *reinterpret_cast<const f32x4*>(R0 + aoff[0][0] + co);
            raw[c][1] = *reinterpret_cast<const f32x4*>(R0 + aoff[0][1] + co);
            raw[c][2] = *reinterpret_cast<const f32x4*>(R0 + aoff[1][0] + co);
            raw[c][3] = *reinterpret_cast<const f32x4*>(R0 + aoff[1][1] + co);
        }
        float Rr[4][8];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = raw[c][e >> 2][e & 3], b = raw[c][2 + (e >> 2)][e & 3];
                asm("v_fma_f32 %0, %1, %2, %3" : "=v"(Rr[c][e]) : "v"(b), "v"(sb), "v"(a));
            }
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            float v[4][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ch = 2 * pr + e;
                asm("v_sub_f32 %0, %1, %2" : "=v"(v[0][e]) : "v"(Rr[0][ch]), "v"(Rr[2][ch]));
                asm("v_add_f32 %0, %1, %2" : "=v"(v[1][e]) : "v"(Rr[1][ch]), "v"(Rr[2][ch]));
                asm("v_sub_f32 %0, %1, %2" : "=v"(v[2][e]) : "v"(Rr[2][ch]), "v"(Rr[1][ch]));
                asm("v_sub_f32 %0, %1, %2" : "=v"(v[3][e]) : "v"(Rr[1][ch]), "v"(Rr[3][ch]));
            }
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                f16x2 h, l;
                split_pair<MIX>(v[nu][0], v[nu][1], h, l);
                Ah[nu][pr] = __builtin_bit_cast(unsigned, h); Al[nu][pr] = __builtin_bit_cast(unsigned, l);
            }
        }
    };

    f32x16 acc[4][NCT];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][ct][r] = 0.f;

    const int b_lane = xil * 16384 + lane * 16;
    // the 24 MFMAs of an interval with this wave's NP requests spread between them (piece(i): request number i)
    auto mma = [&](int slot, auto np_tag, auto&& piece) {
        constexpr int NP = decltype(np_tag)::value;
        if (FAR_WINO_EXP & 2) {
#pragma unroll
            for (int i = 0; i < NP; ++i) piece(i);
            return;
        }
        // Round 5: the multiplying group runs its interval at raised issue priority, so that on every SIMD the wave that feeds the
        // matrix pipe (and issues the interval's requests) wins the arbitration against its transforming partner: -2 ... -3.4 % on every
        // shape (profiles/r05_k17_ab.txt; the transforming group at raised priority: +3 %; priority 3 instead of 2: no better)
        if (p.prio) __builtin_amdgcn_s_setprio(2);
        const unsigned char* B = Bs + slot * SLAB + b_lane;
        f16x8 bh[2][NCT], bl[2][NCT];
        auto read_b = [&](int nu) {
            const int q = nu & 1;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                bh[q][ct] = *reinterpret_cast<const f16x8*>(B + (nu * 4 + 2 * ct) * 1024);
                bl[q][ct] = *reinterpret_cast<const f16x8*>(B + (nu * 4 + 2 * ct + 1) * 1024);
            }
        };
        read_b(0);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const int q = nu & 1;
            if (nu + 1 < 4) read_b(nu + 1);
#pragma unroll
            for (int m = 0; m < 3 * NCT; ++m) {
                const int ct = HALF ? 0 : m & 1, term = HALF ? m : m >> 1;       // hi.hi, hi.lo, lo.hi
                const f16x8 a = __builtin_bit_cast(f16x8, term < 2 ? Ah[nu] : Al[nu]);
                const f16x8 bb = term == 1 ? bl[q][ct] : bh[q][ct];
                acc[nu][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, acc[nu][ct], 0, 0, 0);
                const int mi = 3 * NCT * nu + m;
                bool any = false;
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if ((HALF ? (i * 12) / (NP > 12 ? NP : 12) : (i < 8 ? i : 9 + 2 * (i - 8))) == mi) { piece(i); any = true; }
                if (any) __builtin_amdgcn_sched_barrier(0);        // the request stays behind this MFMA
            }
        }
        if (p.prio) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: slabs 0, 1, raw patches 0, 1; the xi = 0, 1 waves transform k-step 0
    {
        const unsigned char* s0 = slab_src(Q ? 1 : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) b_piece(s0, Q ? 1 : 0, j);
#pragma unroll
        for (int j = 0; j < NRP; ++j) raw_piece(0, 0, j);
#pragma unroll
        for (int j = 0; j < NRP; ++j) raw_piece(1, 1, j);
    }
    FAR_WINO_STAMP(0);
    if (FAR_WINO_EXP & 128) asm volatile("s_barrier" ::: "memory");      // experiment: what the exposed prologue latency costs
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (FAR_WINO_EXP & 1) transform(-1);
    if (!Q) transform(0);
    FAR_WINO_STAMP(1);

    // ---- K loop: two intervals per k-step.  Interval 2k: waves 0-3 multiply k-step k (slab 2k) and request slab 2k+2, then raw
    // patch k+2; waves 4-7 transform k-step k.  Interval 2k+1: waves 4-7 multiply (slab 2k+1) and request slab 2k+3; waves 0-3
    // transform k-step k+1.  Slab i sits in ring slot i % 3 (slab i+3 is requested in interval i+1, after its last reader), raw
    // patch k in slot k % 3 (patch k+3 is requested in interval 2k+2; patch k is last read in interval 2k).
    // Waits (vmcnt retires in order; a wave never waits for a request younger than one interval):
    //   waves 0-3, end of a multiplying interval: vmcnt(14) -- all but this interval's 8 + 6 requests: the raw patch requested
    //     two intervals ago has landed, one interval before its first reader starts;
    //   waves 0-3, end of a transforming interval: vmcnt(6) -- the slab requested in the interval before, which is read next;
    //   waves 4-7, end of a transforming interval: vmcnt(0) -- their slab of the interval before; multiplying: none.
    int slot_e = 0, slot_o = 1;                     // ring slots of slabs 2k, 2k+1
    int rs0 = 0, rs1 = 1, rs2 = 2;                  // ring slots of raw patches k, k+1, k+2
    for (int k = 0; k < nk; ++k) {
        const int slot_n = 3 - slot_e - slot_o;     // the third slot: slab 2k+2
        if (!Q) {
            const unsigned char* sn = slab_src(2 * k + 2);
            mma(slot_e, std::integral_constant<int, 14>{}, [&](int i) {
                if (i < 8) b_piece(sn, slot_n, i);
                else raw_piece(k + 2, rs2, i - 8);
            });
        } else {
            transform(rs0);
        }
        __builtin_amdgcn_sched_barrier(0);          // the interval's work stays on this side of the barrier
        if (k < 14) FAR_WINO_STAMP(2 + 4 * k);
        FAR_WINO_T2(k, 1);
#ifdef FAR_WINO_TIMING2
        if (Q) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else if (wskip) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        FAR_WINO_T2(k, 2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
        if (Q || (FAR_WINO_EXP & 32)) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (wskip) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // this interval's six raw requests only
        else asm volatile("s_waitcnt vmcnt(14)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        if (k < 14) FAR_WINO_STAMP(3 + 4 * k);
        FAR_WINO_T2(k, 3);
        __builtin_amdgcn_sched_barrier(0);
        if (Q) {
            const unsigned char* sn = slab_src(2 * k + 3);
            mma(slot_o, std::integral_constant<int, 8>{}, [&](int i) { b_piece(sn, slot_e, i); });       // -> the slot slab 2k just left
        } else if (k + 1 < nk) {
            transform(rs1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k < 14) FAR_WINO_STAMP(4 + 4 * k);
        FAR_WINO_T2(k, 4);
#ifdef FAR_WINO_TIMING2
        if (!Q) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        FAR_WINO_T2(k, 5);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
        if (FAR_WINO_EXP & 32) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (!Q) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        if (k < 14) FAR_WINO_STAMP(5 + 4 * k);
        FAR_WINO_T2(k, 6);
        FAR_WINO_T2(k + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        slot_o = slot_e;                            // slab 2k+3 took the slot slab 2k left
        slot_e = slot_n;                            // slab 2k+2
        const int t0 = rs0;
        rs0 = rs1; rs1 = rs2; rs2 = t0;             // raw patch k+3 takes the slot patch k left
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // trailing requests landed before LDS is reused
#ifdef FAR_WINO_TIMING2
    if (blockIdx.x < 4096 && (threadIdx.x & 63) < 16)
        g_wino_stamps2[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (threadIdx.x & 63)] =
            *reinterpret_cast<volatile unsigned long long*>(smem + 162816 + (threadIdx.x >> 6) * 128 + (threadIdx.x & 63) * 8);
#endif
    FAR_WINO_STAMP(60);
    if (FAR_WINO_EXP & 16) {
        float tsum = 0.f;
        for (int nu = 0; nu < 4; ++nu) for (int ct = 0; ct < NCT; ++ct) for (int r = 0; r < 16; ++r) tsum += acc[nu][ct][r];
        if (tsum == 123.456f) p.y[0] = tsum;
        return;
    }

    // ---- activation-range guard (as K9): a finite accumulator set cannot overflow its own sum
    if (p.overflow) {
        float chk = 0.f;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) chk += acc[nu][ct][r];
        const bool bad = !(fabsf(chk) <= FLT_MAX);
        if (__any(bad) && lane == 0) atomicOr(p.overflow, 1);
    }

    // ---- output transform.  In-lane over nu: Z[j] = sum_nu A^T[j][nu] M[nu], written to LDS as the image [xi][j][tile 64][co 64]
    // (a lane holds output channel 32 ct + l31 of the tiles 32 tb + mfma row (r, h): 128-byte runs per store instruction and
    // half-wave).  Then wave w turns tile row w into pixels: lane (pixel of a group of four, channel quad q) reads the four Z(xi)
    // of its (tile, column j) as 16-byte vectors, Y[0] = Z0 + Z1 + Z2, Y[1] = Z1 - Z2 - Z3, and stores both output rows with
    // 16-byte stores (256 contiguous bytes per pixel and workgroup).
    float* const zb = reinterpret_cast<float*>(smem);
    {
        float* const zw = zb + ((xi * 2) * 64 + 32 * tb + 4 * h) * 64 + l31;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)          // (HALF: the image's second channel tile keeps stale bytes; its lanes store nothing)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 8 * (r >> 2) + (r & 3);                         // + 4 h: the tile inside the block
                zw[m * 64 + 32 * ct] = (acc[0][ct][r] + acc[1][ct][r]) + acc[2][ct][r];
                zw[64 * 64 + m * 64 + 32 * ct] = (acc[1][ct][r] - acc[2][ct][r]) - acc[3][ct][r];
            }
    }
    const int q = lane & 15, pi = lane >> 4;
    const int co4 = cb * 64 + 4 * q;
    const bool cok = co4 < p.Cout;                   // Cout % 4 == 0: a quad is in or out as a whole
    const float as = p.act == 2 ? p.slope : 0.f, ab = p.act == 0 ? -__builtin_inff() : 0.f;     // act(v) = max(v, v as + ab)
    const int oy = oy0 + 2 * wave;                   // output rows oy, oy + 1 of tile row `wave`
    const float* __restrict__ resp = p.res;
    float* __restrict__ yp = p.y;
    // residual rows first: they arrive behind the barrier and the LDS reads
    f32x4 rr[4][2];
    bool ok[4][2];
    long off[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ox = ox0 + 4 * it + pi;
            ok[it][i] = cok && (oy + i < p.H) && (ox < p.W);
            off[it][i] = (((long)img * p.H + oy + i) * p.W + ox) * p.Cout + co4;
            // unconditional loads (no branch: the compiler then counts the requests exactly, and none is pending on a skipped
            // path when the stores start -- every wait between stores would also wait for the stores before it): rows outside
            // the image and launches without a residual read the zero row
            rr[it][i] = *reinterpret_cast<const f32x4*>((resp && ok[it][i]) ? resp + off[it][i] : p.zeros);
        }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    FAR_WINO_STAMP(61);
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(rr[it][i]));         // the compiler's wait for the loads sits here, once
    sc4 = sc4 * p.out_mul;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int x = 4 * it + pi;                   // pixel column inside the block: tile column x >> 1, output column j = x & 1
        const float* zr = zb + (((x & 1) * 64) + 8 * wave + (x >> 1)) * 64 + 4 * q;
        const f32x4 z0 = *reinterpret_cast<const f32x4*>(zr);
        const f32x4 z1 = *reinterpret_cast<const f32x4*>(zr + 2 * 64 * 64);
        const f32x4 z2 = *reinterpret_cast<const f32x4*>(zr + 4 * 64 * 64);
        const f32x4 z3 = *reinterpret_cast<const f32x4*>(zr + 6 * 64 * 64);
        f32x4 y0 = (z0 + z1) + z2, y1 = (z1 - z2) - z3;
        y0 = y0 * sc4 + sh4 + rr[it][0];
        y1 = y1 * sc4 + sh4 + rr[it][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y0[e] = fmaxf(y0[e], __builtin_fmaf(y0[e], as, ab));
            y1[e] = fmaxf(y1[e], __builtin_fmaf(y1[e], as, ab));
        }
        if (ok[it][0]) *reinterpret_cast<f32x4*>(yp + off[it][0]) = y0;
        if (ok[it][1]) *reinterpret_cast<f32x4*>(yp + off[it][1]) = y1;
    }
#ifdef FAR_WINO_TIMING
    FAR_WINO_STAMP(62);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FAR_WINO_STAMP(63);
#endif
}

template <bool MIX>
__global__ __launch_bounds__(512, 2) void k_wino(const WinoArgs p) {
    // the workgroup's channel block (as wino_body maps it): the last one may hold at most 32 channels
    const long seq = (p.ntb & 7) == 0 ? (long)blockIdx.x >> 3 : (long)blockIdx.x;
    const int cb = (int)(seq % p.ncb);
    const bool half = p.half_ok && cb == p.ncb - 1;            // workgroup-uniform
    if (half) {
        if (threadIdx.x >= 256) wino_body<true, MIX, true>(p);
        else wino_body<false, MIX, true>(p);
    } else {
        if (threadIdx.x >= 256) wino_body<true, MIX, false>(p);
        else wino_body<false, MIX, false>(p);
    }
}

// Packs w (read through element strides like K9's pack: s_co, s_ci per channel, s_tap per tap in execution order ky * 3 + kx)
// into the image [co block][k-step][half][xi & 1][nu][co tile][plane][lane][8] fp16: lane = (co = 64 cb + 32 ct + (lane & 31),
// k-group lane >> 5), element e = input channel 16 k + 8 (lane >> 5) + e; value = sign(xi) * (G g G^T)[xi][nu] * wmul in float64,
// hi = fp16(value), lo = fp16(value - hi); sign = -1 for xi = 2 (the kernel's transform yields -V there).
__global__ __launch_bounds__(256) void k_wino_pack(const float* __restrict__ w, long s_co, long s_ci, long s_tap, int Cin, int Cout,
                                                   int nk, int ncb, const float* __restrict__ wmul_dev, _Float16* __restrict__ out,
                                                   const float* __restrict__ base_scale, float* __restrict__ scale_vec) {
    const float wmul = wmul_dev[0];
    if (scale_vec && blockIdx.x == 0)
        for (int co = threadIdx.x; co < Cout; co += blockDim.x) scale_vec[co] = (base_scale ? base_scale[co] : 1.0f) * wmul_dev[1];
    const long total = (long)ncb * nk * 2 * 32 * 64;           // 16-byte items
    if (blockIdx.x == 0 && threadIdx.x < 16) out[(size_t)total * 8 + threadIdx.x] = (_Float16)0.f;      // the zero row
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int lane = (int)(t & 63); t >>= 6;
        const int plane = (int)(t & 1); t >>= 1;
        const int ct = (int)(t & 1); t >>= 1;
        const int nu = (int)(t & 3); t >>= 2;
        const int xl = (int)(t & 1); t >>= 1;
        const int half = (int)(t & 1); t >>= 1;
        const int k = (int)(t % nk);
        const int cb = (int)(t / nk);
        const int xi = 2 * half + xl;
        const int co = cb * 64 + ct * 32 + (lane & 31);
        _Float16* dst = out + (size_t)i * 8;
        for (int e = 0; e < 8; ++e) {
            const int ci = 16 * k + 8 * (lane >> 5) + e;
            double u = 0.0;
            if (co < Cout && ci < Cin) {
                const float* g = w + (long)co * s_co + (long)ci * s_ci;
                for (int a = 0; a < 3; ++a) {
                    double row = 0.0;
                    for (int b = 0; b < 3; ++b) row += (double)g[(long)(3 * a + b) * s_tap] * G[nu][b];
                    u += G[xi][a] * row;
                }
                u *= (double)wmul;
                if (xi == 2) u = -u;
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

// Bytes of the Winograd image of a [Cout][Cin][3][3] weight (hi + lo planes, + the zero row padding lanes read).
size_t far_wino_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    const size_t nk = (Cin + 15) / 16, ncb = (Cout + 63) / 64;
    return ncb * nk * 2 * SLAB + 32;
}

// w: the element of tap 0 of a [Cout][Cin][3][3] weight read through element strides (a contiguous torch weight: 9 Cin, 9, 1);
// scale_in = { 2^w_exp, 2^-(w_exp + 4) } on the device (far_weight_scale_f32: 2^13 <= max|w| 2^w_exp < 2^14, so that
// |G g G^T| 2^w_exp <= 2.25 * 2^14 stays in fp16); scale_vec_out[co] = base_scale[co] (1 when NULL) * scale_in[1] (may be NULL).
int far_wino_pack_view_scaled_f32(const float* w, long s_co, long s_ci, long s_tap, int Cin, int Cout, const float* scale_in,
                                  void* packed, const float* base_scale, float* scale_vec_out, hipStream_t stream) {
    far_clear_errors();
    if (!w || !packed || !scale_in || far_wino_packed_bytes(Cin, Cout) == 0) return FAR_EINVAL;
    const int nk = (Cin + 15) / 16, ncb = (Cout + 63) / 64;
    const long items = (long)ncb * nk * 2 * 32 * 64;
    long blocks = (items + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(k_wino_pack, dim3((unsigned)blocks), dim3(256), 0, stream, w, s_co, s_ci, s_tap, Cin, Cout, nk, ncb, scale_in,
                       (_Float16*)packed, base_scale, scale_vec_out);
    return far_check_launch();
}

// y = act(scale[co] * conv3x3(x, W)[co] + shift[co] + res), stride 1, zero padding 1, NHWC fp32 -- far_conv_nhwc_f32's contract for
// ksize = 3, stride = 1, split = 1 with `packed` a far_wino_pack_* image; fields the Winograd kernel does not serve (x2, ln_*,
// post_res, up, out_planes > 1, res_group > 1, act_scale_dev, act_exp < 0) are rejected with FAR_EINVAL.  Cin % 4 == 0.
// The activations are split unscaled: |a| <= 16376; `scale` is corrected by 2^4 for the 2^-4 it folds.
int far_conv3x3_wino_f32(const far_conv_desc* desc, hipStream_t stream) {
    if (!desc) return FAR_EINVAL;
    const far_conv_desc& d = *desc;
    far_clear_errors();
    if (d.N == 0) return FAR_OK;
    if (!d.x || !d.packed || !d.scale || !d.y || d.N < 0 || d.H <= 0 || d.W <= 0 || d.Cin <= 0 || (d.Cin & 3) || d.Cout <= 0 ||
        d.ksize != 3 || d.stride != 1 || !d.split || d.act < 0 || d.act > 2 || (d.act == 2 && !(d.slope >= 0.f && d.slope <= 1.f)) ||
        d.x == d.y || d.x2 || d.Cin1 != d.Cin || d.ln_gamma || d.ln_beta || d.post_res || d.up || d.out_planes != 1 || d.res_group != 1 ||
        d.act_scale_dev || d.act_exp < 0 || d.act_exp > 8 || (d.Cout & 3))
        return FAR_EINVAL;
    WinoArgs a;
    a.x = d.x; a.w = (const unsigned char*)d.packed;
    a.zeros = reinterpret_cast<const float*>((const unsigned char*)d.packed + far_wino_packed_bytes(d.Cin, d.Cout) - 32);
    a.scale = d.scale; a.shift = d.shift; a.res = d.res; a.y = d.y; a.overflow = d.overflow;
    a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Cout = d.Cout; a.nk = (d.Cin + 15) / 16; a.ncb = (d.Cout + 63) / 64;
    a.tilesX = (d.W + 15) / 16; a.tilesY = (d.H + 15) / 16;
    a.ntb = d.N * a.tilesX * a.tilesY;
    a.act = d.act; a.slope = d.slope; a.out_mul = 16.0f;
    a.half_ok = (d.Cout - 64 * (a.ncb - 1) <= 32 && far_get_tuning(9) == 0) ? 1 : 0;      // tuning 9: 1 = every block on the full body
    a.prio = far_get_tuning(15) == 0 ? 1 : 0;                   // tuning 15 = 1: no priority change (the round-4 kernel)
    const long nblk = a.ntb * a.ncb;
    if (nblk > 0x7fffffffL || (long)d.H * d.W * d.Cout > 0x7fffffffL) return FAR_EINVAL;
    const bool mix = far_get_tuning(8) == 0;
    bool cfg_failed = false;
    FAR_ONCE_PER_DEVICE(cfg_failed = hipFuncSetAttribute((const void*)k_wino<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess ||
                                     hipFuncSetAttribute((const void*)k_wino<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess);
    if (cfg_failed) return far_check_launch();
    if (mix) hipLaunchKernelGGL(k_wino<true>, dim3((unsigned)nblk), dim3(512), SMEM, stream, a);
    else hipLaunchKernelGGL(k_wino<false>, dim3((unsigned)nblk), dim3(512), SMEM, stream, a);
    return far_check_launch();
}

#ifdef FAR_WINO_TIMING2
int far_wino_timing2_dump(void* host, int nblocks) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_stamps2), (size_t)nblocks * 8 * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -5;
}
#endif
#ifdef FAR_WINO_TIMING
int far_wino_timing_dump(void* host, int nblocks) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_stamps), (size_t)nblocks * 8 * 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -5;
}
#endif

}  // extern "C"
