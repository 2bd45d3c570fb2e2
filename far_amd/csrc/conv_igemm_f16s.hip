// K9: implicit-GEMM convolution / linear layer on the f16 matrix cores with SPLIT-PRECISION operands
// (fp32 tensors in and out, fp32 accumulation), with the inference BatchNorm / bias, the activation and the residual
// add fused into the epilogue.
//
// Replaces, for inference, the 3x3 and 1x1 stride-1 convolutions of the ResNet-FPN backbone
//   mp3d_loftr/src/loftr/backbone/resnet_fpn.py:5-12   (conv1x1 / conv3x3)
//                                               :15-43 (BasicBlock: conv -> bn -> relu -> conv -> bn -> +x -> relu)
//                                               :101-119 (ResNetFPN_8_2.forward: layer*_outconv, layer*_outconv2)
// and nn.Linear layers (a 1x1 convolution over a [rows][K] matrix).  These are 55 % (convolutions) + 17 % (linear
// layers) of the fp32 step on the vendor libraries, whose fp32 kernels cannot use the 16x faster f16 matrix rate.
//
// Numerics.  Every fp32 operand v is scaled by a power of two and split as v = hi + lo with hi = fp16(v),
// lo = fp16(v - hi): 22 significand bits, |v - hi - lo| <= 2^-24 |v| (down to the fp16 subnormal floor, which the
// scaling keeps ~2^-29 of the activation scale).  The product uses three MFMAs, hi.hi + hi.lo + lo.hi (each f16 x f16
// product is exact in fp32; the dropped lo.lo term is <= 2^-24 |a||b|), accumulated in fp32 by the matrix core:
// fp32-grade results (measured against a float64 convolution in tests/test_conv_gpu.py, next to the vendor fp32
// Winograd kernel's error) at 16/3 of the fp32 MFMA rate.  SPLIT = false keeps only hi (plain fp16 operands).
//
// Tiling (gfx950).  Workgroup = 4 waves, every wave owns 64 output pixels (4 tile rows x 16 columns; 64 consecutive
// rows for 1x1 / linear) x 128 output channels = 2 x 4 accumulator tiles of 32x32x16 f16 MFMAs (128 registers).
// Cout <= 128: 4 x 1 waves, WG tile 256 pixels x 128 channels; wider: 2 x 2 waves, 128 pixels x 256 channels.
// K loop: input-channel chunks of 32; inside a chunk the KS*KS taps; inside a tap two 16-channel k-steps = "phases".
//  * pixels: per chunk the (TH+2) x (16+2) input halo is read ONCE from HBM (fp32, requested three taps early into
//    registers), split and written to LDS after the chunk's last phase; all 9 taps read it at shifted addresses
//    (80-byte pixel stride, 256-byte-aligned rows: conflict-free ds_read_b128 fragment reads).
//  * weights: per phase one NT x 16-channel slab (pre-split and pre-swizzled by far_conv_pack_f32 in execution order,
//    so the global image IS the LDS image and the source pointer just advances) streams in by LDS-DMA
//    (global_load_lds_dwordx4) into a ring of three slabs, two phases ahead (counted vmcnt + raw s_barrier: the
//    barrier never drains the DMA queue).  One barrier per phase.
//  * per phase and wave: 12 ds_read_b128 fragment reads (the pixel fragments one phase ahead, the weight fragments in
//    two halves, the second behind the first half's MFMAs) and 24 MFMAs (split: hi.hi + hi.lo + lo.hi of one 16-channel
//    k-step) / 16 (plain operands: one phase per tap, both k-steps of the chunk, the slab's two planes = the two
//    k-steps).  The phase body is one basic block (unconditional slab request, taps and k-steps unrolled) whose issue order is pinned with
//    sched_group_barrier: one MFMA, then the LDS reads / DMA requests / address arithmetic that fit in the issue
//    slots its 32-cycle pass leaves free (+5-7 % over letting them queue up in front of the MFMA block).
//  * memory waits: loads, LDS-DMA requests and stores share ONE in-order vmcnt on gfx9, and with LDS-DMA requests
//    pending the compiler turns every wait on a load into vmcnt(0).  So nothing touches a staged pixel value before
//    stage_store (padding lanes load a zero row appended to the packed weights instead of being masked), the slab
//    waits are explicit and count the pixel loads in flight, and the epilogue's store loops are straight-line with all
//    their loads (residual rows, LayerNorm parameters) marked arrived up front: stores stream back to back.
//  * epilogue: acc * scale + shift through a 16 KiB-per-wave LDS transpose to 16-byte row accesses; optional residual
//    (one row per pixel or per group of rows), fused LayerNorm (+ post-norm residual), or the FPN merge (2x bilinear
//    upsampling of the coarser level added, `up`, its own template instantiation); activation max(v, v a + b).
// Measured on MI355X (64 images, bench.py / tools/k9_ab.py): 3x3 layers 331-438 TFLOP/s fp32-equivalent (1.0-1.3 PFLOP/s
// of f16 MFMA issue: 40-53 % of the matrix peak at the nominal 2.4 GHz, ~75 % of the pipe's cycles at the 1.5-1.8 GHz
// the part sustains under this load, tools/k9_timing.py) including the fused epilogue, against 100-125 TFLOP/s for
// the vendor fp32 Winograd convolution alone.
#include "common.h"
#include "linear_small.h"
#include <vector>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TW = 16;                  // output tile columns of the 3x3 kernels (rows: 4 per pixel-wave)
constexpr int A_PXB = 80;               // LDS bytes per staged pixel per plane: 32 fp16 channels + 16 B pad
constexpr int ACT_EXP_DEFAULT = 4;      // activations are scaled by 2^act_exp before the split (see far_conv_nhwc_f32)

// LDS image of the staged input pixels of one 32-channel chunk.  3x3: the (TH+2) x (16+2) halo, row stride rounded
// up to a multiple of 256 B so that, with the odd 5-slot pixel stride, every 16-lane group of a ds_read_b128
// fragment read (lanes {0-3,12-15,20-27}, ... = 8 pixels of one tile row + 8 of the next) hits 16 distinct slots.
template <int KS, int MW, int ST>
struct Geo {
    static constexpr int TH = 4 * MW;
    static constexpr int HW = ST * (TW - 1) + KS, HH = ST * (TH - 1) + KS;   // input patch of one output tile
    static constexpr int PX = KS == 1 ? 64 * MW : HW * HH;       // staged pixels per chunk
    static constexpr int ROWB = KS == 1 ? 0 : (HW * A_PXB + 255) / 256 * 256;
    static constexpr int A_PLANE = KS == 1 ? PX * A_PXB : HH * ROWB;
    static constexpr int ITEMS = PX * 4;                          // (pixel, 8-channel group) staging items
    __device__ static constexpr int px_off(int hy, int hx) { return hy * ROWB + hx * A_PXB; }
};

struct ConvArgs {
    const float* x;              // input channels [0, Cin1)
    const float* x2;             // input channels [Cin1, Cin) (fused concatenation), or null when Cin1 == Cin
    int Cin1;
    const unsigned char* w;      // packed weights (far_conv_pack_f32)
    const float* zeros;          // the 32 zero bytes that end the packed image
    const float* scale;          // [Cout] multiplies the accumulator (BN scale / 1, with the operand scaling folded in)
    const float* shift;          // [Cout] or null
    const float* res;            // residual, same layout as y, or null
    int res_group;               // > 1: res is [npix / res_group][Cout], row pix / res_group is added to pixel pix
    const float* ln_gamma;       // fused LayerNorm over the Cout channels of every pixel (after res / act), or null
    const float* ln_beta;
    float ln_eps;
    const float* post_res;       // added after the LayerNorm (y's layout), or null
    const float* up;             // [N][Ho/2][Wo/2][Cout]: its bilinear 2x upsampling (align_corners) is added (1x1 mode), or null
    float* y;
    long npix;                   // N * Ho * Wo output pixels (= input pixels for 1x1)
    long ntiles;                 // blocks along x
    int Csub;                    // output channels per output plane (Cout: one NHWC tensor)
    int H, W, Ho, Wo, Cin, Cout, tilesX, tilesY, nchunks, nblkY, act;   // input H x W, output Ho x Wo
    float slope;
    float act_scale;             // 2^act_exp: activations are multiplied by it before the fp16 split (default 2^4)
    float out_mul;               // 2^(4 - act_exp): `scale` folds 2^-4, this corrects it for the scale actually used
    int* overflow;               // device flag, |= 1 when an accumulator of this launch is not finite (may be null)
    const float* scale_dev;      // device { act_scale, out_mul } overriding the two fields above (may be null)
    float* kv_part;              // far_linear_kv_f16s (the EPI = 1 instantiations): the columns are the k | v projections of a LoFTR
    int kv_S, kv_valid;          // layer, head-interleaved, and the epilogue is la_kv_epilogue (K'^T V partial sums), not a store
    const unsigned char* kv_img; // far_linear_q_apply_f16s (EPI = 2): per image the K'^T V state as MFMA operands + ksum (KV_IMG_BYTES)
    float la_eps, la_vlen;       // (EPI = 2: kv_S = tokens per image on the query side, la_vlen = the source's length S)
    // EPI 1 / 2, image lengths that are no multiple of 64: the launch runs on a PADDED geometry -- kv_S = the length rounded up to 64
    // rows per image, kv_valid = the real length (= kv_S otherwise) -- so that a 64-row block never holds rows of two images; the
    // rows behind an image's end read nothing (row table, as the gather mode) and are masked / not stored
    int sub2;                    // 1x1 mode on every second pixel of every second row of x (the stride-2 shortcut, resnet_fpn.py:26-29)
    const long* g_b;             // far_linear_gather_f16s (EPI = 3): row r of the launch is token r % (W W) of window r / (W W), read
    const long* g_cell;          // straight from the fine map x [n][Hf][Wf][Cin] at the window's position (fine_preprocess.py:40-47)
    int g_wc, g_W, g_stride, g_Hf, g_Wf;
};

__device__ __forceinline__ void split8(const float4& u, const float4& v, float act_scale, f16x8& hi, f16x8& lo) {
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 8; i += 2) {                       // two channels per packed conversion (common.h: split2)
        f16x2 h, l;
        split2(f32x2{x[i], x[i + 1]} * f32x2{act_scale, act_scale}, h, l);
        hi[i] = h.x; hi[i + 1] = h.y;
        lo[i] = l.x; lo[i + 1] = l.y;
    }
}

// Register staging of one chunk's input pixels: item i = (pixel i >> 2, 8-channel group i & 3).
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ITERS>
struct Stage {
    f32x4 v[ITERS][2];
};

struct TilePos {
    int img, oy0, ox0;
    long pix0;
};

// 3x3: which input pixel (index into [N][H][W], -1 = outside the image / no item) each of this thread's staging items reads.  The same
// for every chunk of the K loop: worked out once per workgroup (a division, four bounds tests and a 64-bit multiply-add per item and
// chunk otherwise -- ~45 VALU instructions x ITERS in the phase that requests the pixels).
template <int KS, int MW, int ST, int NTHR, int ITERS>
__device__ __forceinline__ void stage_pixels(int (&pixv)[ITERS], const ConvArgs& p, const TilePos& tp, int tid) {
    using G = Geo<KS, MW, ST>;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = tid + NTHR * it;
        const int hp = i >> 2;
        const int hy = hp / G::HW, hx = hp - hy * G::HW;
        const int iy = ST * tp.oy0 - KS / 2 + hy, ix = ST * tp.ox0 - KS / 2 + hx;
        const bool ok = i < G::ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        pixv[it] = ok ? (tp.img * p.H + iy) * p.W + ix : -1;          // N H W < 2^31 (checked by the host)
    }
}

template <int KS, int MW, int ST, int NTHR, int ITERS>
__device__ __forceinline__ void stage_load(Stage<ITERS>& st, const ConvArgs& p, const TilePos& tp, int chunk, int tid, bool live = true,
                                           const int* rowpix = nullptr, const int* pixv = nullptr) {
    using G = Geo<KS, MW, ST>;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = tid + NTHR * it;
        const int hp = i >> 2, g = i & 3;
        const int c0 = 32 * chunk + 8 * g;
        bool ok = live && i < G::ITEMS;          // !live: the request is still issued (no branch), to the zero row
        long pix;
        if (KS != 1 && pixv) {
            pix = pixv[it];
            ok = live && pixv[it] >= 0;
        } else if (KS == 1) {
            pix = tp.pix0 + hp;
            ok = ok && pix < p.npix;
            if (rowpix) {                        // gather mode: the input pixel of each row of the tile, from the workgroup's LDS table
                const int gp = rowpix[hp < 64 * MW ? hp : 0];
                pix = gp;
                ok = ok && gp >= 0;
            } else if (p.sub2) {                 // output pixel (n, oy, ox) reads input pixel (n, 2 oy, 2 ox): no subsampled copy of x
                const unsigned up = (unsigned)pix, rowg = up / (unsigned)p.Wo, ox = up - rowg * (unsigned)p.Wo;
                const unsigned n = rowg / (unsigned)p.Ho, oy = rowg - n * (unsigned)p.Ho;
                pix = ((long)n * p.H + 2 * oy) * p.W + 2 * ox;
            }
        } else {
            const int hy = hp / G::HW, hx = hp - hy * G::HW;
            const int iy = ST * tp.oy0 - KS / 2 + hy, ix = ST * tp.ox0 - KS / 2 + hx;
            ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            pix = ((long)tp.img * p.H + iy) * p.W + ix;
        }
        // unconditional 16-byte loads: padding (outside the image / past Cin) reads the 32 zero bytes that end the
        // packed weight image (a kernel-argument pointer like x: plain global loads) instead of being masked afterwards -- a select on the loaded value would make the wave wait for HBM right here instead of
        // at stage_store, one chunk later
        const bool ok0 = ok && c0 < p.Cin, ok1 = ok && c0 + 4 < p.Cin;
#if defined(FAR_K9_EXP) && (FAR_K9_EXP & 4)      // experiment: real data, but always the same 1024 pixels (cache-resident)
        pix &= 1023;
#endif
        const bool second = c0 >= p.Cin1;                       // Cin1 % 8 == 0: a group never straddles the inputs
        const float* base = second ? p.x2 + pix * (p.Cin - p.Cin1) + (c0 - p.Cin1) : p.x + pix * p.Cin1 + c0;
#if defined(FAR_K9_EXP) && (FAR_K9_EXP & 1)      // experiment (never in the product build): no activation traffic
        const float* src0 = p.zeros;
        const float* src1 = p.zeros;
        (void)base; (void)ok0; (void)ok1;
#else
        const float* src0 = ok0 ? base : p.zeros;
        const float* src1 = ok1 ? base + 4 : p.zeros;
#endif
        // The loads are issued from `asm`: a load the compiler knows about makes it wait for the staged registers with vmcnt(0) at the
        // chunk's end (with LDS-DMA requests pending it never counts), which drains the weight slabs requested since -- a full memory
        // round trip exposed per chunk (every 2 phases in Linear mode).  stage_arrived waits for exactly these loads instead.
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st.v[it][0]) : "v"(src0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st.v[it][1]) : "v"(src1) : "memory");
    }
}

// Waits until the staged values have arrived: stage_load's requests are older than the YOUNGER weight-slab requests of the chunk
// (the memory counter retires in order), so vmcnt(YOUNGER) is exact and leaves those slabs in flight.  The staged registers are
// operands of the statement: nothing that reads them can be scheduled above it, and nothing touches them between the asm loads
// and this wait (tools/k9_asm_check.py scans the generated code for exactly that).
template <int YOUNGER, int ITERS>
__device__ __forceinline__ void stage_arrived(Stage<ITERS>& st) {
    static_assert(YOUNGER >= 0 && YOUNGER < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(YOUNGER) : "memory");
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        f32x4 &u = st.v[it][0], &w = st.v[it][1];
        asm volatile("" : "+v"(u), "+v"(w));
    }
}

template <int KS, int MW, int ST, int NTHR, int ITERS, bool SPLIT>
__device__ __forceinline__ void stage_store(const Stage<ITERS>& st, unsigned char* As, int tid, float act_scale) {
    using G = Geo<KS, MW, ST>;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = tid + NTHR * it;
        const float4 u = make_float4(st.v[it][0].x, st.v[it][0].y, st.v[it][0].z, st.v[it][0].w);
        const float4 w = make_float4(st.v[it][1].x, st.v[it][1].y, st.v[it][1].z, st.v[it][1].w);
        if (ITERS * NTHR == G::ITEMS || i < G::ITEMS) {
            const int hp = i >> 2, g = i & 3;
            int off;
            if (KS == 1) off = hp * A_PXB;
            else { const int hy = hp / G::HW; off = G::px_off(hy, hp - hy * G::HW); }
            f16x8 hi, lo;
            split8(u, w, act_scale, hi, lo);
            *reinterpret_cast<f16x8*>(As + off + g * 16) = hi;
            if (SPLIT) *reinterpret_cast<f16x8*>(As + G::A_PLANE + off + g * 16) = lo;
        }
    }
}

// Epilogue: y = act(acc * scale[c] + shift[c] + res).  A lane holds output channel
// cout_w + 32 nt + l31 of 16 pixels per accumulator tile; the residual values of a tile are all requested before the
// first store.  Output channel co lands in plane co / Csub of y ([Cout / Csub][pixels][Csub]; Csub = Cout: plain NHWC).
template <int KS, int NTW>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, const f32x16 (&acc)[2][NTW], const TilePos& tp,
                                              int cout_w, int wm, int l31, int h) {
    const float* __restrict__ resp = p.res;
    float* __restrict__ yp = p.y;
    // Accumulator register r of a lane covers tile pixel  row (r >> 3), column (r & 3) + 8 ((r >> 2) & 1) + 4 h  of
    // its 2 x 16 pixel MFMA tile (32 consecutive pixels in GEMM mode): 32-bit offsets from one 64-bit base per tile.
    const int rowstep = (KS == 1 ? 16 : p.Wo) * p.Csub;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int mtile = 2 * wm + mt;
        long pix0;                       // first pixel of the tile
        int nrow, ncol;                  // valid tile rows / columns
        if (KS == 1) {
            pix0 = tp.pix0 + 32 * mtile;
            const long left = p.npix - pix0;
            nrow = left >= 32 ? 2 : (left > 16 ? 2 : (left > 0 ? 1 : 0));
            ncol = left >= 32 ? 16 : -1;                               // -1: ragged, test the linear index instead
        } else {
            const int oy = tp.oy0 + 2 * mtile;
            pix0 = ((long)tp.img * p.Ho + oy) * p.Wo + tp.ox0;
            nrow = p.Ho - oy;
            ncol = p.Wo - tp.ox0;
        }
        const long left = KS == 1 ? p.npix - pix0 : 0;
        int delta[16];
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int trow = r >> 3, tcol = (r & 3) + 8 * ((r >> 2) & 1) + 4 * h;
            delta[r] = trow * rowstep + tcol * p.Csub;
            ok[r] = (KS == 1 && ncol < 0) ? (16 * trow + tcol < left) : (trow < nrow && tcol < ncol);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int co = cout_w + 32 * nt + l31;
            if (co >= p.Cout) continue;
            const float sc = p.scale[co] * p.out_mul, sh = p.shift ? p.shift[co] : 0.f;
            const int plane = co / p.Csub;
            const long base = (long)plane * p.npix * p.Csub + pix0 * p.Csub + (co - plane * p.Csub);
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                rv[r] = 0.f;
                if (resp && ok[r]) {
                    if (p.res_group > 1) {        // one residual row per group of consecutive pixels (GEMM mode only)
                        const long pix = pix0 + 16 * (r >> 3) + ((r & 3) + 8 * ((r >> 2) & 1) + 4 * h);
                        rv[r] = resp[(pix / p.res_group) * p.Cout + co];
                    } else {
                        rv[r] = resp[base + delta[r]];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][nt][r] * sc + sh + rv[r];
                if (p.act == 1) v = fmaxf(v, 0.f);
                else if (p.act == 2) v = v > 0.f ? v : v * p.slope;
                if (ok[r]) yp[base + delta[r]] = v;
            }
        }
    }
}

// 16-byte output stores of the wide epilogue.  FAR_K9_EXP bit 3 (experiment build only): non-temporal stores.
#if defined(FAR_K9_EXP) && (FAR_K9_EXP & 8)
#define FAR_K9_STORE4(ptr, val) __builtin_nontemporal_store(f32x4{(val).x, (val).y, (val).z, (val).w}, reinterpret_cast<f32x4*>(ptr))
#else
#define FAR_K9_STORE4(ptr, val) (*reinterpret_cast<float4*>(ptr) = (val))
#endif

#ifdef FAR_K9_TIMING
// Development instrumentation (tools/k9_timing.py; never defined in the product build): per-workgroup s_memtime
// stamps at kernel entry, after the prologue, after the K loop and at exit, plus the hardware id of wave 0.
__device__ unsigned long long g_k9_stamps[12 * 65536];
#define FAR_K9_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) g_k9_stamps[12 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FAR_K9_STAMP(i) do {} while (0)
#endif

// Sum over the 32 lanes of each half-wave, result in every lane: DPP butterflies inside the 16-lane rows, one
// cross-row exchange.
#define FAR_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float sum32(float v) {
    v += FAR_DPP_F(v, 0xB1);       // quad_perm [1,0,3,2]
    v += FAR_DPP_F(v, 0x4E);       // quad_perm [2,3,0,1]
    v += FAR_DPP_F(v, 0x141);      // row_half_mirror
    v += FAR_DPP_F(v, 0x140);      // row_mirror
    return v + shfl_xor_f(v, 16);
}

// "Uses" a loaded value (no instruction) so that the compiler's wait for it sits here, on every path: consumption
// inside an exec-masked block otherwise leaves the load pending on the skip path, and every later merge point gets a
// conservative s_waitcnt vmcnt(0) -- which on gfx9 also waits for all stores issued so far.
__device__ __forceinline__ void arrived(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// Wide epilogue (Cout, Csub multiples of 4): every 32-pixel x 128-channel accumulator tile of a wave goes through a
// 16 KiB LDS transpose so that a lane ends up with 4 consecutive channels of one pixel: 16-byte residual loads and
// 16-byte stores, each store instruction writing two whole 512-byte pixel rows (4x fewer memory instructions than
// the per-register path above).  `lw` = this wave's private 32 x 128 float region; the caller has synchronised the
// workgroup (the main loop's LDS buffers are dead).
template <int KS, int NW, int UP>
__device__ __forceinline__ void conv_epilogue_wide(const ConvArgs& p, const f32x16 (&acc)[2][4], const TilePos& tp,
                                                   int cout_w, int wm, int wave, int lane, float* lw, float* xch,
                                                   int mask_mt = -1, int mask_lo = 0, int mask_hi = 0) {
    // mask_mt >= 0 (N7 mode): in pixel tile mask_mt the lanes mask_lo <= l31 < mask_hi (four channels each) hold a tile this
    // wave did not compute: they neither load nor store
    const int l31 = lane & 31, h = lane >> 5;
    const float* __restrict__ resp = p.res;
    float* __restrict__ yp = p.y;
    float sc[4], sh[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int co = cout_w + 32 * nt + l31;
        sc[nt] = co < p.Cout ? p.scale[co] * p.out_mul : 0.f;
        sh[nt] = (co < p.Cout && p.shift) ? p.shift[co] : 0.f;
    }
    const int co4 = cout_w + 4 * l31;                      // this lane's 4 channels in the store phase
    const bool cok_all = co4 < p.Cout;
    const bool lane_masked = l31 >= mask_lo && l31 < mask_hi;
    const int plane = cok_all ? co4 / p.Csub : 0;
    const long cbase = (long)plane * p.npix * p.Csub + (co4 - plane * p.Csub);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const bool cok = cok_all && !(mt == mask_mt && lane_masked);
        const int mtile = 2 * wm + mt;
        long pix0;
        int nrow, ncol;
        long left = 0;
        if (KS == 1) {
            pix0 = tp.pix0 + 32 * mtile;
            left = p.npix - pix0;
            nrow = 2; ncol = 16;
        } else {
            const int oy = tp.oy0 + 2 * mtile;
            pix0 = ((long)tp.img * p.Ho + oy) * p.Wo + tp.ox0;
            nrow = p.Ho - oy;
            ncol = p.Wo - tp.ox0;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = 16 * (r >> 3) + (r & 3) + 8 * ((r >> 2) & 1) + 4 * h;       // pixel of the tile
                lw[q * 128 + 32 * nt + l31] = acc[mt][nt][r] * sc[nt] + sh[nt];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same wave: LDS executes its requests in order
        if (mt == 0) FAR_K9_STAMP(6);
        if (p.ln_gamma) {
            // ---- fused LayerNorm (transformer.py:61, 65-67): the whole channel row of a pixel is in this wave's tile
            // (Cout = 128) or in the tiles of the two waves that share its pixels (Cout = 256, partial sums exchanged
            // through `xch`): two-pass mean / variance like K6, then gamma, beta and the post-norm residual.
            // The tile stays in LDS and is re-read for each of the three passes (sum, centred squares, output): only
            // the per-pixel statistics and the residual rows are live across the reductions.
            float part[16], mean[16];
            // the post-norm residual rows are requested first: they arrive behind the reductions below
            float4 pr[16];
            if (p.post_res) {
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int q = 2 * it + h;
                    const int trow = q >> 4, tcol = q & 15;
                    const bool ok = cok && (KS == 1 ? (q < left) : (trow < nrow && tcol < ncol));
                    const long off = (pix0 + (KS == 1 ? q : trow * p.Wo + tcol)) * p.Csub + cbase;
                    pr[it] = *reinterpret_cast<const float4*>(p.post_res + (ok ? off : 0));
                }
            } else {
#pragma unroll
                for (int it = 0; it < 16; ++it) pr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4 g4 = *reinterpret_cast<const float4*>(p.ln_gamma + co4), b4 = *reinterpret_cast<const float4*>(p.ln_beta + co4);
            const float inv_c = 1.0f / (float)p.Cout;
            auto all_rows = [&](float (&x)[16]) {               // sum over the channel row of every pixel
#pragma unroll
                for (int it = 0; it < 16; ++it) x[it] = sum32(x[it]);
                if (NW == 2) {
                    __syncthreads();                             // previous exchange consumed
                    if (l31 == 0) {
#pragma unroll
                        for (int it = 0; it < 16; ++it) xch[wave * 32 + 2 * it + h] = x[it];
                    }
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < 16; ++it) x[it] += xch[(wave ^ 1) * 32 + 2 * it + h];
                }
            };
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const float4 v = *reinterpret_cast<const float4*>(lw + (2 * it + h) * 128 + 4 * l31);
                part[it] = (v.x + v.y) + (v.z + v.w);
            }
            all_rows(part);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                float4 v = *reinterpret_cast<const float4*>(lw + (2 * it + h) * 128 + 4 * l31);
                mean[it] = part[it] * inv_c;
                v.x -= mean[it]; v.y -= mean[it]; v.z -= mean[it]; v.w -= mean[it];
                part[it] = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            all_rows(part);
            if (mt == 0) FAR_K9_STAMP(7);
            arrived(g4); arrived(b4);
#pragma unroll
            for (int it = 0; it < 16; ++it) arrived(pr[it]);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int q = 2 * it + h;
                const int trow = q >> 4, tcol = q & 15;
                const bool ok = cok && (KS == 1 ? (q < left) : (trow < nrow && tcol < ncol));
                const long off = (pix0 + (KS == 1 ? q : trow * p.Wo + tcol)) * p.Csub + cbase;
                const float rstd = 1.0f / sqrtf(part[it] * inv_c + p.ln_eps);
                float4 v = *reinterpret_cast<const float4*>(lw + q * 128 + 4 * l31);
                v.x -= mean[it]; v.y -= mean[it]; v.z -= mean[it]; v.w -= mean[it];
                float4 o;
                o.x = v.x * rstd * g4.x + b4.x + pr[it].x; o.y = v.y * rstd * g4.y + b4.y + pr[it].y;
                o.z = v.z * rstd * g4.z + b4.z + pr[it].z; o.w = v.w * rstd * g4.w + b4.w + pr[it].w;
                if (ok) FAR_K9_STORE4(yp + off, o);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (mt == 0) FAR_K9_STAMP(5);
            continue;
        }
        // Plain path.  All residual rows of the tile are requested first, then two batches of 8 rows: LDS reads,
        // activation, stores.  Straight-line code (no branch: the compiler then counts the outstanding requests
        // exactly) and no wait that has to see a store acknowledged: on gfx9 stores count in vmcnt too, and a residual
        // load issued after a store cannot be waited for without waiting for that store (~600 cycles each when loads
        // and stores alternate, which is what made the epilogue 9-34 % of a workgroup's lifetime).  Per-lane 64-bit
        // base pointers once per tile, 32-bit row offsets after that.
        // Activation as max(v, v * as + ab): ReLU (0, 0), LeakyReLU (slope in [0, 1], 0), none (0, -inf).
        const float as = p.act == 2 ? p.slope : 0.f, ab = p.act == 0 ? -__builtin_inff() : 0.f;
        float* const ybase = yp + (pix0 * (long)p.Csub + cbase);
        const int leftc = left > 64 ? 64 : (int)left;
        auto row_ok = [&](int it) { return cok && (KS == 1 ? (2 * it + h < leftc) : ((it >> 3) < nrow && 2 * (it & 7) + h < ncol)); };
        auto row_off = [&](int it) { return (unsigned)((KS == 1 ? 2 * it + h : (it >> 3) * p.Wo + 2 * (it & 7) + h) * p.Csub); };
        auto tile_out = [&](auto has_res) {
            constexpr bool HR = decltype(has_res)::value;
            float4 rr[16];
            if (HR) {
                const bool grouped = p.res_group > 1;               // fine_preprocess: one residual row per res_group rows
                const unsigned G = grouped ? (unsigned)p.res_group : 1u;
                const long g0 = pix0 / (long)G;
                const unsigned rem0 = (unsigned)(pix0 - g0 * (long)G);
                const float* rbase = grouped ? resp + (g0 * (long)p.Cout + co4) : resp + (pix0 * (long)p.Csub + cbase);
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const unsigned ro = grouped ? ((rem0 + (unsigned)(2 * it + h)) / G) * (unsigned)p.Cout : row_off(it);
                    rr[it] = *reinterpret_cast<const float4*>(row_ok(it) ? rbase + ro : resp);     // always a valid address
                }
#pragma unroll
                for (int it = 0; it < 16; ++it) arrived(rr[it]);
            }
#pragma unroll
            for (int b0 = 0; b0 < 16; b0 += 8) {
                float4 vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) vv[j] = *reinterpret_cast<const float4*>(lw + (2 * (b0 + j) + h) * 128 + 4 * l31);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float4 v = vv[j];
                    if (HR) { v.x += rr[b0 + j].x; v.y += rr[b0 + j].y; v.z += rr[b0 + j].z; v.w += rr[b0 + j].w; }
                    v.x = fmaxf(v.x, __builtin_fmaf(v.x, as, ab)); v.y = fmaxf(v.y, __builtin_fmaf(v.y, as, ab));
                    v.z = fmaxf(v.z, __builtin_fmaf(v.z, as, ab)); v.w = fmaxf(v.w, __builtin_fmaf(v.w, as, ab));
                    float* dst = ybase + row_off(b0 + j);
                    if (row_ok(b0 + j)) FAR_K9_STORE4(dst, v);
                }
            }
        };
        // Fused FPN merge (resnet_fpn.py:108-109, :113-114): y = conv1x1(x) + bilinear_2x(up), align_corners = True; the
        // arithmetic of K8 (k_upsample2x_add: torch's source index and hy * (hx a + lx b) + ly * (hx c + lx d)), so the
        // result is bit-identical to the two-kernel sequence without writing and re-reading the lateral tensor.
        auto tile_up = [&]() {
            constexpr int UB = 2;                                      // rows per batch and lane half (register budget)
            const int hc = p.Ho >> 1, wc = p.Wo >> 1;
            const float ry = p.Ho > 1 ? (float)(hc - 1) / (float)(p.Ho - 1) : 0.f;
            const float rx = p.Wo > 1 ? (float)(wc - 1) / (float)(p.Wo - 1) : 0.f;
            const long per = (long)p.Ho * p.Wo;
            const long n0 = pix0 / per;                                // tile start -> (image, row, column); Wo >= 32:
            const int rem = (int)(pix0 - n0 * per);                    // a tile row wraps at most once
            const int Y0 = rem / p.Wo, X0 = rem - Y0 * p.Wo;
#pragma unroll
            for (int b0 = 0; b0 < 16; b0 += UB) {
                float4 ta[UB], tb[UB], tc[UB], td[UB], vv[UB];
                float wy[UB], wx[UB];
#pragma unroll
                for (int j = 0; j < UB; ++j) {
                    const int q = 2 * (b0 + j) + h;
                    int X = X0 + q, Y = Y0;
                    long n = n0;
                    if (X >= p.Wo) { X -= p.Wo; ++Y; }
                    if (Y >= p.Ho) { Y = 0; ++n; }
                    const float sy = ry * (float)Y, sx = rx * (float)X;
                    const int y0 = (int)sy, x0 = (int)sx;
                    const int y1 = y0 + (y0 < hc - 1 ? 1 : 0), x1 = x0 + (x0 < wc - 1 ? 1 : 0);
                    wy[j] = sy - (float)y0; wx[j] = sx - (float)x0;
                    const bool ok = row_ok(b0 + j);
                    const float* base = p.up + (ok ? (n * hc * (long)wc) * p.Cout + co4 : 0);
                    const int o00 = ok ? (y0 * wc + x0) * p.Cout : 0, o01 = ok ? (y0 * wc + x1) * p.Cout : 0;
                    const int o10 = ok ? (y1 * wc + x0) * p.Cout : 0, o11 = ok ? (y1 * wc + x1) * p.Cout : 0;
                    ta[j] = *reinterpret_cast<const float4*>(base + o00); tb[j] = *reinterpret_cast<const float4*>(base + o01);
                    tc[j] = *reinterpret_cast<const float4*>(base + o10); td[j] = *reinterpret_cast<const float4*>(base + o11);
                }
#pragma unroll
                for (int j = 0; j < UB; ++j) vv[j] = *reinterpret_cast<const float4*>(lw + (2 * (b0 + j) + h) * 128 + 4 * l31);
#pragma unroll
                for (int j = 0; j < UB; ++j) { arrived(ta[j]); arrived(tb[j]); arrived(tc[j]); arrived(td[j]); }
#pragma unroll
                for (int j = 0; j < UB; ++j) {
                    const float ly = wy[j], lx = wx[j], hy = 1.f - ly, hx = 1.f - lx;
                    float4 v = vv[j];
                    v.x += hy * (hx * ta[j].x + lx * tb[j].x) + ly * (hx * tc[j].x + lx * td[j].x);
                    v.y += hy * (hx * ta[j].y + lx * tb[j].y) + ly * (hx * tc[j].y + lx * td[j].y);
                    v.z += hy * (hx * ta[j].z + lx * tb[j].z) + ly * (hx * tc[j].z + lx * td[j].z);
                    v.w += hy * (hx * ta[j].w + lx * tb[j].w) + ly * (hx * tc[j].w + lx * td[j].w);
                    v.x = fmaxf(v.x, __builtin_fmaf(v.x, as, ab)); v.y = fmaxf(v.y, __builtin_fmaf(v.y, as, ab));
                    v.z = fmaxf(v.z, __builtin_fmaf(v.z, as, ab)); v.w = fmaxf(v.w, __builtin_fmaf(v.w, as, ab));
                    float* dst = ybase + row_off(b0 + j);
                    if (row_ok(b0 + j)) FAR_K9_STORE4(dst, v);
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the batches apart (hoisting all 16 rows' addresses spills)
            }
        };
        // The same merge when the tile's 32 pixels lie in one image row (Wo % 32 == 0: the backbone's 320- and 160-pixel rows): the
        // source rows y0 / y1 and the vertical weights are the wave's, and a lane walks 16 CONSECUTIVE pixels, whose source columns
        // advance by 0 or 1 per pixel -- the left pair of a pixel is the previous pixel's left or right pair, so only the right pair
        // is loaded: 34 instead of 64 16-byte gathers per lane and tile, the next four pixels' requests in flight while four are
        // blended (the generic form drains its 8 requests every 2 pixels: the epilogue, not the 128-channel main loop, was this
        // launch -- 2.87 ms against 1.59 ms for the same convolution without `up`).  Same values, same formula: bit-identical.
        auto tile_up_row = [&]() {
            constexpr int PB = 4;                                      // pixels per batch and lane half (8 spills)
            const int hc = p.Ho >> 1, wc = p.Wo >> 1;
            const float ry = p.Ho > 1 ? (float)(hc - 1) / (float)(p.Ho - 1) : 0.f;
            const float rx = p.Wo > 1 ? (float)(wc - 1) / (float)(p.Wo - 1) : 0.f;
            const long per = (long)p.Ho * p.Wo;
            const long n0 = pix0 / per;
            const int rem = (int)(pix0 - n0 * per);
            const int Y0 = rem / p.Wo, Xh = rem - Y0 * p.Wo + 16 * h;
            const float sy = ry * (float)Y0;
            const int y0 = (int)sy, y1 = y0 + (y0 < hc - 1 ? 1 : 0);
            const float ly = sy - (float)y0, hy = 1.f - ly;
            const float* top = p.up + (cok ? ((n0 * hc + y0) * (long)wc) * p.Cout + co4 : 0);
            const float* bot = p.up + (cok ? ((n0 * hc + y1) * (long)wc) * p.Cout + co4 : 0);
            const int cstep = cok ? p.Cout : 0;
            auto xsrc = [&](int jj, int& x0, int& x1, float& lx) {
                const float sx = rx * (float)(Xh + jj);
                x0 = (int)sx; x1 = x0 + (x0 < wc - 1 ? 1 : 0); lx = sx - (float)x0;
            };
            float4 T[2][PB], B[2][PB];
            auto issue = [&](int b, float4 (&t)[PB], float4 (&bb)[PB]) {
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    int x0, x1; float lx;
                    xsrc(b * PB + j, x0, x1, lx);
                    t[j] = *reinterpret_cast<const float4*>(top + x1 * cstep);
                    bb[j] = *reinterpret_cast<const float4*>(bot + x1 * cstep);
                }
            };
            int xprev; float lx0; int x1u;
            xsrc(0, xprev, x1u, lx0);
            float4 a = *reinterpret_cast<const float4*>(top + xprev * cstep), c = *reinterpret_cast<const float4*>(bot + xprev * cstep);
            float4 tprev = a, bprev = c;
            issue(0, T[0], B[0]);
            float* const yrow = ybase + (long)(16 * h) * p.Csub;
#pragma unroll
            for (int b = 0; b < 16 / PB; ++b) {
                if (b + 1 < 16 / PB) issue(b + 1, T[(b + 1) & 1], B[(b + 1) & 1]);
                float4 vv[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) vv[j] = *reinterpret_cast<const float4*>(lw + (16 * h + b * PB + j) * 128 + 4 * l31);
                if (b == 0) { arrived(a); arrived(c); }
#pragma unroll
                for (int j = 0; j < PB; ++j) { arrived(T[b & 1][j]); arrived(B[b & 1][j]); }
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    int x0, x1; float lx;
                    xsrc(b * PB + j, x0, x1, lx);
                    const bool adv = x0 != xprev;                      // the source column moved on: left pair = the previous right pair
                    a.x = adv ? tprev.x : a.x; a.y = adv ? tprev.y : a.y; a.z = adv ? tprev.z : a.z; a.w = adv ? tprev.w : a.w;
                    c.x = adv ? bprev.x : c.x; c.y = adv ? bprev.y : c.y; c.z = adv ? bprev.z : c.z; c.w = adv ? bprev.w : c.w;
                    const float4 tb = T[b & 1][j], td = B[b & 1][j];
                    const float hx = 1.f - lx;
                    float4 v = vv[j];
                    v.x += hy * (hx * a.x + lx * tb.x) + ly * (hx * c.x + lx * td.x);
                    v.y += hy * (hx * a.y + lx * tb.y) + ly * (hx * c.y + lx * td.y);
                    v.z += hy * (hx * a.z + lx * tb.z) + ly * (hx * c.z + lx * td.z);
                    v.w += hy * (hx * a.w + lx * tb.w) + ly * (hx * c.w + lx * td.w);
                    v.x = fmaxf(v.x, __builtin_fmaf(v.x, as, ab)); v.y = fmaxf(v.y, __builtin_fmaf(v.y, as, ab));
                    v.z = fmaxf(v.z, __builtin_fmaf(v.z, as, ab)); v.w = fmaxf(v.w, __builtin_fmaf(v.w, as, ab));
                    if (cok) FAR_K9_STORE4(yrow + (long)(b * PB + j) * p.Csub, v);
                    tprev = tb; bprev = td; xprev = x0;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (UP == 2) tile_up_row();               // separate instantiations: their registers must not weigh on the others
        else if (UP) tile_up();
        else if (resp) tile_out(std::true_type{});
        else tile_out(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // tile read back before the next one overwrites it
        if (mt == 0) FAR_K9_STAMP(5);
    }
}

// ---- The K'^T V state of LinearAttention (linear_attention.py:38-45) straight from the accumulators of the k | v projection
// (far_linear_kv_f16s): k and v never reach HBM.  The packed weight holds the two projections head by head -- columns
// 64 j + [0, 32) = k of head j, 64 j + [32, 64) = v of head j -- so a wave's four 32-column tiles are (k, v) of two heads for its
// 64 rows, and an accumulator register holds ONE row for the lane's channel in both.  Eight registers of a lane (with the
// eight of its lane + 32 twin: 16 rows) therefore ARE an MFMA operand [channel][k = row] of
//     KV[d][v] += K'[row][d] * V[row][v]           (x 1 / S at the end: linear_attention.py:43 divides V first, same value to an ulp)
// -- A from the k tile, B from the v tile, no data movement; products as everywhere: hi * hi + hi * lo + lo * hi of fp16 pairs
// (operands x act_scale like the activations of the main loop: the same range, the same overflow flag).  ksum[d] = sum of K' is an
// in-lane sum.  One partial sum per 64-row block of the launch (the rows of a wave), written to part[block][H D][D + 1];
// k_kv_blocks_reduce adds the blocks of an image in block order: deterministic, no atomics, and independent of how many images
// share the launch (image lengths that are no multiple of 64 run on a padded geometry, ConvArgs: a block never holds two images).
// F.elu(x) + 1 = x + 1 (x > 0), e^x otherwise, as 2^(x log2 e) on the hardware exponential: the rounding of the argument costs
// |x| e^x 2^-24 <= 2^-25 in absolute terms (half an ulp of 1.0, the size of a K'), where expm1f(x) + 1 rounds twice and, like expf,
// costs a dozen instructions per value (64 feature maps per lane in this epilogue: measured 349 -> 330 us per launch).
__device__ __forceinline__ float la_elu1(float x) { return x > 0.f ? x + 1.f : __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

template <bool FULL>
__device__ __forceinline__ void la_kv_pass(const ConvArgs& p, const f32x16 (&acc)[2][4], const float (&sc)[4], const float (&sh)[4],
                                           int lo, int hi, long blk, int head0, int HD, int l31, int h, float act_scale) {
    const float unscale = 1.0f / (act_scale * act_scale), fS = (float)p.kv_valid;
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
        f32x16 kv;
#pragma unroll
        for (int r = 0; r < 16; ++r) kv[r] = 0.f;
        float ks = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float ka[8], vb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = 8 * g + i;
                    ka[i] = la_elu1(acc[mt][2 * hd][r] * sc[2 * hd] + sh[2 * hd]);
                    vb[i] = acc[mt][2 * hd + 1][r] * sc[2 * hd + 1] + sh[2 * hd + 1];
                    if (!FULL) {                         // the rows behind the image's end (zero inputs: K' = 1 there, V = its bias)
                        const int q = 32 * mt + 16 * (r >> 3) + (r & 3) + 8 * ((r >> 2) & 1) + 4 * h;          // row of the wave
                        ka[i] = (q >= lo && q < hi) ? ka[i] : 0.f;
                    }
                    ks += ka[i];
                }
                f16x8 ah, al, bh, bl;
                split8(make_float4(ka[0], ka[1], ka[2], ka[3]), make_float4(ka[4], ka[5], ka[6], ka[7]), act_scale, ah, al);
                split8(make_float4(vb[0], vb[1], vb[2], vb[3]), make_float4(vb[4], vb[5], vb[6], vb[7]), act_scale, bh, bl);
                kv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, kv, 0, 0, 0);
                kv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, kv, 0, 0, 0);
                kv = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, kv, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);       // one 16-row group at a time: hoisting all feature maps spills
            }
        ks += shfl_xor_f(ks, 32);
        if (p.overflow) {                                  // a K' or V beyond the split's range (as the main loop's guard)
            float chk = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) chk += kv[r];
            if (__any(!(fabsf(chk) <= FLT_MAX)) && (l31 | h) == 0) atomicOr(p.overflow, 1);
        }
        float* o = p.kv_part + ((size_t)blk * HD + (head0 + hd) * 32) * 33;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = (kv[r] * unscale) / fS;
        if (h == 0) o[l31 * 33 + 32] = ks;
    }
}

__device__ __forceinline__ void la_kv_epilogue(const ConvArgs& p, const f32x16 (&acc)[2][4], long pixw, int cout_w, int lane,
                                               float act_scale) {
    const int l31 = lane & 31, h = lane >> 5;
    if (pixw >= p.npix) return;                          // (npix = images x kv_S: whole 64-row blocks)
    const long img = pixw / p.kv_S;
    const long vl = (long)p.kv_valid - (pixw - img * p.kv_S);    // rows [0, valid) of the wave are tokens of image `img`
    const int valid = vl > 64 ? 64 : (int)vl;
    const int HD = p.Cout >> 1, head0 = cout_w >> 6;
    float sc[4], sh[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int co = cout_w + 32 * nt + l31;
        sc[nt] = p.scale[co] * p.out_mul;
        sh[nt] = p.shift ? p.shift[co] : 0.f;
    }
    const long blk = pixw >> 6;
    if (valid == 64) la_kv_pass<true>(p, acc, sc, sh, 0, 64, blk, head0, HD, l31, h, act_scale);     // wave-uniform
    else la_kv_pass<false>(p, acc, sc, sh, 0, valid, blk, head0, HD, l31, h, act_scale);             // the image's last block
}

// kv[n][HD][33] = the partial sums of image n's 64-row blocks (Sp / 64 of them, Sp = the padded image length), in block order.
// img (optional, HD = 256): the same state as the operands of
// la_apply_epilogue -- per image [head][k-step u][plane hi | lo][lane][8] fp16 = KV[head][d = 16 u + 8 (lane >> 5) + i][v = lane & 31]
// x act_scale, then ksum [256] fp32 (KV_IMG_BYTES per image).
constexpr int KV_IMG_BYTES = 8 * 2 * 2 * 1024 + 256 * 4;

__global__ __launch_bounds__(256) void k_kv_blocks_reduce(const float* __restrict__ part, int Sp, int per,
                                                          float* __restrict__ kv, unsigned char* __restrict__ img, float act_scale) {
    const int n = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    const int nb = Sp >> 6;
    float s = 0.f;
    const float* src = part + (size_t)n * nb * per + e;
#pragma unroll 8
    for (int g = 0; g < nb; ++g, src += per) s += *src;
    kv[(size_t)n * per + e] = s;
    if (img) {
        unsigned char* im = img + (size_t)n * KV_IMG_BYTES;
        const int dg = e / 33, col = e - dg * 33;
        if (col == 32) {
            reinterpret_cast<float*>(im + 32768)[dg] = s;
        } else {
            const int head = dg >> 5, d = dg & 31, u = d >> 4, hh = (d >> 3) & 1, i = d & 7, ln = col + 32 * hh;
            const float x = s * act_scale;
            const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
            _Float16* o = reinterpret_cast<_Float16*>(im + ((head * 2 + u) * 2) * 1024 + ln * 16) + i;
            o[0] = hi;
            o[512] = lo;                                   // the lo plane: 1 KiB further
        }
    }
}

// ---- LinearAttention's second half (linear_attention.py:46-50) in the epilogue of the q projection (far_linear_q_apply_f16s): the
// launch writes the attention message, q never reaches HBM and K5's apply launch (a read of q, a write of the message) disappears.
// Per 32-row x 128-channel accumulator tile of a wave (4 heads): Q' = elu(q) + 1 goes to the wave's LDS tile in [row][channel]
// order (row stride LDW = 132 floats: the 16-byte reads below are conflict free); lane (row, h) reads its row back as the A operand
// of  num[row][v] = sum_d Q'[row][d] KV[d][v]  (B = the image k_kv_blocks_reduce wrote; products hi * hi + hi * lo + lo * hi),
// forms den = Q'[row] . ksum from the same registers (an fma chain + one cross-lane add), passes Z = 1 / (den + eps) to the lanes that
// hold the row's outputs through the tile's four spare columns, and the message (num Z) S replaces Q' in the tile; 16-byte stores.
constexpr int LDW = 132;

__device__ __forceinline__ void la_apply_epilogue(const ConvArgs& p, const f32x16 (&acc)[2][4], long pixw, int cout_w, int lane,
                                                  float* lw, float act_scale) {
    const int l31 = lane & 31, h = lane >> 5;
    const int Lp = p.kv_S, L = p.kv_valid;               // rows per image in the launch's geometry / real tokens per image
    const float unscale = 1.0f / (act_scale * act_scale), fS = p.la_vlen;
    const int head0 = cout_w >> 5;
    float sc[4], sh[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int co = cout_w + 32 * nt + l31;
        sc[nt] = p.scale[co] * p.out_mul;
        sh[nt] = p.shift ? p.shift[co] : 0.f;
    }
    float chk = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long pix0 = pixw + 32 * mt;
        if (pix0 >= p.npix) break;
        const long im = pix0 / Lp;
        const long t0 = pix0 - im * Lp, leftl = (long)L - t0;      // the tile's rows t0 .. of image `im`; the ones behind L are padding
        if (leftl <= 0) continue;
        const int left = leftl > 32 ? 32 : (int)leftl;
        const unsigned char* img = p.kv_img + (size_t)im * KV_IMG_BYTES;
        const float* ksum = reinterpret_cast<const float*>(img + 32768);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = 16 * (r >> 3) + (r & 3) + 8 * ((r >> 2) & 1) + 4 * h;                 // pixel of the tile
                lw[q * LDW + 32 * nt + l31] = la_elu1(acc[mt][nt][r] * sc[nt] + sh[nt]);
            }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x16 num;
#pragma unroll
            for (int r = 0; r < 16; ++r) num[r] = 0.f;
            float den = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned char* bsrc = img + (((head0 + nt) * 2 + u) * 2) * 1024 + lane * 16;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(bsrc), bl = *reinterpret_cast<const f16x8*>(bsrc + 1024);
                const float* kp = ksum + 32 * (head0 + nt) + 16 * u + 8 * h;
                const float4 k0 = *reinterpret_cast<const float4*>(kp), k1 = *reinterpret_cast<const float4*>(kp + 4);
                const float* ap = lw + l31 * LDW + 32 * nt + 16 * u + 8 * h;
                const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
                den = fmaf(a0.x, k0.x, den); den = fmaf(a0.y, k0.y, den); den = fmaf(a0.z, k0.z, den); den = fmaf(a0.w, k0.w, den);
                den = fmaf(a1.x, k1.x, den); den = fmaf(a1.y, k1.y, den); den = fmaf(a1.z, k1.z, den); den = fmaf(a1.w, k1.w, den);
                f16x8 ah, al;
                split8(a0, a1, act_scale, ah, al);
                num = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, num, 0, 0, 0);
                num = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, num, 0, 0, 0);
                num = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, num, 0, 0, 0);
            }
            den += shfl_xor_f(den, 32);
            if (h == 0) lw[l31 * LDW + 128 + nt] = 1.0f / (den + p.la_eps);                           // linear_attention.py:46
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
                chk += num[r];
                lw[m * LDW + 32 * nt + l31] = ((num[r] * unscale) * lw[m * LDW + 128 + nt]) * fS;   // :50
            }
            __builtin_amdgcn_sched_barrier(0);             // head by head: the tile's columns are reused in place
        }
        float* const ybase = p.y + ((im * L + t0) * (long)p.Csub + cout_w + 4 * l31);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int q = 2 * it + h;
            const float4 v = *reinterpret_cast<const float4*>(lw + q * LDW + 4 * l31);
            if (q < left) FAR_K9_STORE4(ybase + (long)q * p.Csub, v);
        }
    }
    if (p.overflow && __any(!(fabsf(chk) <= FLT_MAX)) && lane == 0) atomicOr(p.overflow, 1);
}

// Workgroup = MW x NW waves: wave (wm, wn) owns 64 pixels (4 tile rows x 16) x 32 NTW output channels.
// N7 (2 x 2 wave layout, 192 < Cout <= 224, e.g. the 196-channel layers): the workgroup covers SEVEN 32-channel tiles
// instead of eight.  Wave column wn = 0 owns tiles 0..3, wn = 1 tiles 3..6; the shared tile 3 is computed by wn = 0 for its
// first 32-pixel tile (mt = 0) and by wn = 1 for its second (mt = 1), so that every wave issues 7 of the 8 MFMA triples of a
// k-step: 12.5 % fewer MFMAs, evenly over the four SIMDs (dropping the all-padding eighth tile from the wn = 1 waves alone
// leaves the SIMDs of the wn = 0 waves as the bottleneck: measured in round 1, -2 %).
template <int KS, int MW, int NW, int NTW, bool SPLIT, int ST, int UP, bool N7, int WN, int EPI = 0>
__device__ __forceinline__ void conv_body(const ConvArgs& p) {
    using G = Geo<KS, MW, ST>;
    constexpr int NTHR = 64 * MW * NW;
    constexpr int PLANES = SPLIT ? 2 : 1;
    constexpr int TAPS = KS * KS;
    constexpr int NT = 32 * NTW * NW;                // output channels per workgroup
    constexpr int A_BUF = PLANES * G::A_PLANE;
    // Plain operands (SPLIT = false): one phase per tap covers the whole 32-channel chunk (two MFMA k-steps), the slab
    // holds the two k-steps where the split slab holds the hi and lo planes: same LDS image size, same DMA pattern,
    // 16 instead of 8 MFMAs per barrier.
    constexpr int B_PLANE = NT * 32;                 // one 16-channel k-step of the weight slab
    constexpr int B_BUF = 2 * B_PLANE;               // split: hi | lo of one k-step; plain: k-step 0 | k-step 1
    constexpr int B_ITERS = (B_BUF + NTHR * 16 - 1) / (NTHR * 16);   // DMA rounds per slab (the last may be partial)
    static_assert(B_BUF % (NTHR * 16) == 0, "stage_arrived<N * B_ITERS> counts exactly B_ITERS LDS-DMA requests per wave and slab: a partial last round (wave-dependent) would make the counted vmcnt waits wrong");
    constexpr int ITERS = (G::ITEMS + NTHR - 1) / NTHR;
    constexpr int LOAD_TAP = TAPS >= 3 ? TAPS - 3 : 0;   // the next chunk's pixels are requested this early
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + A_BUF;          // ring of 3 slabs

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NW, wn = wave % NW, l31 = lane & 31, h = lane >> 5;
    // the activation scale: a launch argument, or two device floats written by far_grad_scale_f32 (gradients: the scale follows
    // the tensor's maximum, chosen on the device without a host round trip)
    const float act_scale = p.scale_dev ? p.scale_dev[0] : p.act_scale;
    const float out_mul = p.scale_dev ? p.scale_dev[1] : p.out_mul;
    FAR_K9_STAMP(0);
#ifdef FAR_K9_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 65536) g_k9_stamps[12 * blockIdx.x + 8] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- which output tile / output-channel block.  1-D grid of ntiles * nblkY blocks.  Hardware places block b on
    // XCD b % 8: each XCD gets a contiguous range of tiles (the halos shared by neighbouring tiles are re-read from
    // that XCD's L2) and runs the nblkY channel blocks of a tile back to back (they re-read the same pixels).
    long t;
    int by;
    {
        const long b = blockIdx.x;
        long seq = b, tile0 = 0;
        if ((p.ntiles & 7) == 0) { seq = b >> 3; tile0 = (b & 7) * (p.ntiles >> 3); }
        t = tile0 + seq / p.nblkY;
        by = (int)(seq % p.nblkY);
    }
    TilePos tp{0, 0, 0, 0};
    if (KS == 1) {
        tp.pix0 = t * (64 * MW);
    } else {
        const int tx = (int)(t % p.tilesX);
        t /= p.tilesX;
        const int ty = (int)(t % p.tilesY);
        tp.img = (int)(t / p.tilesY);
        tp.oy0 = ty * G::TH;
        tp.ox0 = tx * TW;
    }
    static_assert(!N7 || (NW == 2 && NTW == 4), "N7 is a mode of the 2 x 2 wave layout");
    const int cout_w = by * NT + (N7 ? 96 : 32 * NTW) * wn;          // first output channel of this wave

    // ---- weight slabs, stored in execution order [chunk][tap][k-step][cout block]: the LDS-DMA source pointer of
    // the prefetch just advances by one slab per phase (two when the all-padding last k-step is skipped).
    const int nchunks = p.nchunks;
    const int last_nks = !SPLIT ? 1 : (p.Cin - 32 * (nchunks - 1)) > 16 ? 2 : 1;   // phases per tap of the last chunk
    const size_t slab_stride = (size_t)p.nblkY * B_BUF;
    const unsigned char* wsrc = p.w + (size_t)by * B_BUF + tid * 16;
    // The request is unconditional (no branch in the phase body): past the last slab it re-reads that slab into the
    // ring slot nobody reads any more.  Slabs are stored in execution order; when the all-padding second k-step of
    // the last chunk is skipped, the pointer advances by two slabs there.
    const int nslab = SPLIT ? ((nchunks - 1) * 2 + last_nks) * TAPS : nchunks * TAPS;
    const int first_last = (nchunks - 1) * 2 * TAPS;           // first slab of the last chunk (split)
    int pidx = 0, pslot = 0;
    auto prefetch = [&]() {
        unsigned char* dst = Bs + pslot * B_BUF + wave * 1024;
#pragma unroll
        for (int j = 0; j < B_ITERS; ++j)
#if defined(FAR_K9_EXP) && (FAR_K9_EXP & 32)     // experiment: one slab request per eight (stale weights: wrong results)
            if ((pidx & 7) == 0)
#endif
            if ((j + 1) * NTHR * 16 <= B_BUF || j * NTHR * 16 + wave * 1024 < B_BUF)       // wave-uniform, static
                __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + j * (NTHR * 16)), (lptr_t)(dst + j * (NTHR * 16)), 16, 0, 0);
        pslot = pslot == 2 ? 0 : pslot + 1;
        const size_t adv = (SPLIT && pidx >= first_last && last_nks == 1) ? 2 * slab_stride : slab_stride;
        ++pidx;
        wsrc += pidx < nslab ? adv : 0;
    };

    // ---- per-lane fragment offsets
    int a_off[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int mtile = 2 * wm + mt;
        if (KS == 1) a_off[mt] = (32 * mtile + l31) * A_PXB + h * 16;
        else a_off[mt] = G::px_off(ST * (2 * mtile + (l31 >> 4)), ST * (l31 & 15)) + h * 16;
    }
    // weight row n of the slab: 2 slots of 8 channels, slot ^= (n >> 3) & 1 (conflict-free fragment reads)
    const int b_off = ((N7 ? 96 : 32 * NTW) * wn + l31) * 32 + ((h ^ ((l31 >> 3) & 1)) * 16);

    f32x16 acc[2][NTW];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // ---- gather mode (EPI = 3, far_linear_gather_f16s): the tile's rows are window tokens; where each one lies in the fine map is
    // worked out ONCE per workgroup (one thread per row: two index loads, three divisions) into a table behind the loop's LDS buffers
    // -- inside the epilogue's region, which is only used after the last staging pass
    const int* rowpix = nullptr;
    if constexpr (EPI == 1 || EPI == 2) {
        // the LinearAttention epilogues on image lengths that are no multiple of 64: the launch's rows are images padded to whole
        // 64-row blocks; the same kind of table says which input row (if any) each row of the tile is
        if (p.kv_valid != p.kv_S) {                      // launch-uniform
            int* tab = reinterpret_cast<int*>(smem + A_BUF + 3 * B_BUF);
            if (tid < 64 * MW) {
                const long row = tp.pix0 + tid;
                const long im = row / p.kv_S;
                const int t = (int)(row - im * p.kv_S);
                tab[tid] = (row < p.npix && t < p.kv_valid) ? (int)(im * p.kv_valid + t) : -1;
            }
            __syncthreads();
            rowpix = tab;
        }
    }
    if constexpr (EPI == 3) {
        static_assert(KS == 1 && !UP && !N7, "gather mode is a Linear-mode input form");
        int* tab = reinterpret_cast<int*>(smem + A_BUF + 3 * B_BUF);
        if (tid < 64 * MW) {
            const long row = tp.pix0 + tid;
            int gp = -1;
            if (row < p.npix) {
                const unsigned ww = (unsigned)(p.g_W * p.g_W), urow = (unsigned)row;
                const unsigned w = urow / ww, t = urow - w * ww;
                const unsigned cell = (unsigned)p.g_cell[w], b = (unsigned)p.g_b[w];
                const int cy = (int)(cell / (unsigned)p.g_wc), cx = (int)(cell - (unsigned)cy * (unsigned)p.g_wc);
                const int ky = (int)(t / (unsigned)p.g_W), kx = (int)(t - (unsigned)ky * (unsigned)p.g_W);
                const int y = cy * p.g_stride - p.g_W / 2 + ky, x = cx * p.g_stride - p.g_W / 2 + kx;
                if ((unsigned)y < (unsigned)p.g_Hf && (unsigned)x < (unsigned)p.g_Wf) gp = ((int)b * p.g_Hf + y) * p.g_Wf + x;
            }
            tab[tid] = gp;
        }
        __syncthreads();
        rowpix = tab;
    }

    // ---- prologue: pixels of chunk 0, weight slabs of phases 0 and 1 (the ring runs two phases ahead)
    Stage<ITERS> st;
    int pixv[ITERS];
    if constexpr (KS != 1) stage_pixels<KS, MW, ST, NTHR, ITERS>(pixv, p, tp, tid);
    stage_load<KS, MW, ST, NTHR, ITERS>(st, p, tp, 0, tid, true, rowpix, KS != 1 ? pixv : nullptr);
    prefetch();
    prefetch();
    stage_arrived<2 * B_ITERS>(st);
    stage_store<KS, MW, ST, NTHR, ITERS, SPLIT>(st, As, tid, act_scale);
    FAR_K9_STAMP(1);

    // Fragment registers.  The pixel (A) fragments of a phase are read during the previous phase; the weight (B)
    // fragments in groups of NH column tiles, double buffered: group g + 1 is read behind group g's MFMAs.  Two groups
    // of NTW / 2 tiles normally; single tiles (16 fewer registers) where the staging registers of a large halo
    // (ITERS > 3: the 4 x 1 wave layout of the 3x3 mode) would otherwise spill inside the K loop.
    constexpr bool APRE = ITERS <= 3;      // pixel fragments one phase ahead (needs 16 more registers)
    constexpr int NH = APRE ? NTW / 2 : 1;
    constexpr int NGRP = NTW / NH;
    f16x8 ah[2], al[2], ahn[2], aln[2], bh[2][NH], bl[2][NH];
    auto read_a = [&](f16x8 (&xh)[2], f16x8 (&xl)[2], int tap, int ks) {
        const int tapoff = KS == 1 ? 0 : G::px_off(tap / KS, tap % KS);
        const unsigned char* A = As + tapoff + ks * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            xh[mt] = *reinterpret_cast<const f16x8*>(A + a_off[mt]);
            xl[mt] = *reinterpret_cast<const f16x8*>(A + (SPLIT ? G::A_PLANE : 32) + a_off[mt]);   // lo plane / second k-step
        }
    };
    auto read_b = [&](int grp, const unsigned char* B) {
        const int half = grp & 1;
#pragma unroll
        for (int q = 0; q < NH; ++q) {
#if defined(FAR_K9_EXP) && (FAR_K9_EXP & 16)     // experiment: half the weight-fragment reads (wrong results)
            if (q > 0) { bh[half][q] = bh[half][0]; bl[half][q] = bl[half][0]; continue; }
#endif
            bh[half][q] = *reinterpret_cast<const f16x8*>(B + (grp * NH + q) * 32 * 32);
            bl[half][q] = *reinterpret_cast<const f16x8*>(B + B_PLANE + (grp * NH + q) * 32 * 32);
        }
    };
    // (WN: the wave column as a compile-time constant in N7 mode -- the skipped tile depends on it -- and 0 otherwise)
    auto mma_half = [&](int grp) {
        const int half = grp & 1;
#if defined(FAR_K9_EXP) && (FAR_K9_EXP & 64)     // experiment: no MFMAs (one fma per fragment pair keeps the reads alive; wrong results)
#pragma unroll
        for (int q = 0; q < NH; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt][grp * NH + q][0] += (float)ah[mt][0] * (float)bh[half][q][0] + (float)al[mt][0] * (float)bl[half][q][0];
        return;
#endif
        // N7: wn = 0 leaves out (mt 1, tile 3), wn = 1 leaves out (mt 0, tile 0) -- the half of the shared tile the other column computes
        auto live = [&](int mt, int nt) { return !N7 || (WN == 0 ? !(mt == 1 && nt == 3) : !(mt == 0 && nt == 0)); };
#pragma unroll
        for (int q = 0; q < NH; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                if (live(mt, grp * NH + q))
                    acc[mt][grp * NH + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[half][q], acc[mt][grp * NH + q], 0, 0, 0);
        if (SPLIT) {
#pragma unroll
            for (int q = 0; q < NH; ++q)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    if (live(mt, grp * NH + q))
                        acc[mt][grp * NH + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[half][q], acc[mt][grp * NH + q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NH; ++q)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    if (live(mt, grp * NH + q))
                        acc[mt][grp * NH + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[half][q], acc[mt][grp * NH + q], 0, 0, 0);
        } else {                                   // second k-step of the chunk: its own pixel and weight fragments
#pragma unroll
            for (int q = 0; q < NH; ++q)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    if (live(mt, grp * NH + q))
                        acc[mt][grp * NH + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bl[half][q], acc[mt][grp * NH + q], 0, 0, 0);
        }
    };

    int slot = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int nks = !SPLIT ? 1 : chunk == nchunks - 1 ? last_nks : 2;            // phases per tap
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks >= nks) continue;                                      // between phases; the body is branch free
                const bool first = tap == 0 && ks == 0;                       // first phase of a chunk
                const bool last = tap == TAPS - 1 && ks == nks - 1;           // last phase of a chunk
                // This phase's slab has landed once only the requests issued after it are outstanding (vmcnt retires
                // in order): the younger slab's DMAs, and -- in the two phases after the pixel-load phase -- the next
                // chunk's pixel loads, which were issued behind this slab's DMAs and need not have arrived yet.
                const bool after_load = SPLIT ? (tap == LOAD_TAP && ks == 1) || (TAPS > 1 && tap == LOAD_TAP + 1 && ks == 0)
                                              : TAPS > 1 && (tap == LOAD_TAP + 1 || tap == LOAD_TAP + 2);
                if (after_load) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_ITERS + 2 * ITERS) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_ITERS) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const unsigned char* B = Bs + slot * B_BUF + b_off;
                read_b(0, B);
                if (first || !APRE) read_a(ah, al, tap, ks);
                __builtin_amdgcn_sched_barrier(0);
                prefetch();
                if (tap == LOAD_TAP && ks == 0) __builtin_amdgcn_sched_barrier(0);   // the slab's DMAs stay ahead of the pixel loads (the waits count on it)
                if (tap == LOAD_TAP && ks == 0)            // unconditional (no branch in the body): the last chunk requests zeros
                    stage_load<KS, MW, ST, NTHR, ITERS>(st, p, tp, chunk + 1, tid, chunk + 1 < nchunks, rowpix, KS != 1 ? pixv : nullptr);
#pragma unroll
                for (int grp = 0; grp < NGRP; ++grp) {
                    if (grp + 1 < NGRP) read_b(grp + 1, B);
                    if (grp == NGRP - 1 && APRE && !(tap == TAPS - 1 && (ks == 1 || !SPLIT))) {   // pixel fragments of the next phase of this chunk
                        const bool wrap = ks + 1 >= nks;           // (a skipped second k-step makes the next phase (tap + 1, 0))
                        read_a(ahn, aln, wrap ? tap + 1 : tap, wrap ? 0 : 1);
                    }
                    mma_half(grp);
                }
                // issue order: one MFMA, then the other instructions of the phase (LDS reads, the DMA requests and
                // their address arithmetic) in the issue slots its 32-cycle pass leaves free
#pragma unroll
                for (int i = 0; i < (N7 ? (SPLIT ? 21 : 14) : (SPLIT ? 6 : 4) * NTW); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read (LDS-DMA, pixel loads)
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);      // VALU / SALU
                }
                __builtin_amdgcn_sched_barrier(0);
                (void)last;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) if (APRE) { ah[mt] = ahn[mt]; al[mt] = aln[mt]; }
                slot = slot == 2 ? 0 : slot + 1;
            }
        }
        // slab requests younger than the pixel loads of phase (LOAD_TAP, 0): one per later phase of the chunk (two phases per tap with
        // split operands; a last chunk with one phase per tap has fewer, and its staged values are not used)
        stage_arrived<((SPLIT ? 2 * (TAPS - LOAD_TAP) - 1 : TAPS - LOAD_TAP - 1)) * B_ITERS>(st);
        if (chunk + 1 < nchunks) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with this chunk's pixels
            stage_store<KS, MW, ST, NTHR, ITERS, SPLIT>(st, As, tid, act_scale);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (unused) slab requests must land before LDS is reused
    FAR_K9_STAMP(2);

    // Activation-range guard: an input beyond the fp16 range of the split (|a| 2^act_exp > 65504) becomes inf and reaches
    // the accumulators as inf / NaN.  The sum of a wave's accumulators is not finite exactly then (finite accumulators
    // cannot overflow it: they are sums of products of fp16 numbers); one atomic per wave that saw it, none otherwise.
    if (p.overflow) {
        float chk = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) chk += acc[mt][nt][r];
        const bool bad = !(fabsf(chk) <= FLT_MAX);
        if (__any(bad) && lane == 0) atomicOr(p.overflow, 1);
    }
#if defined(FAR_K9_EXP) && (FAR_K9_EXP & 2)      // experiment: no epilogue (one store keeps the accumulators alive)
    {
        float t = 0.f;
        for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < NTW; ++nt) for (int r = 0; r < 16; ++r) t += acc[mt][nt][r];
        if (t == 123.456f) p.y[0] = t;
        return;
    }
#endif
    ConvArgs pe = p;                 // the epilogues read the output multiplier from their argument block
    pe.out_mul = out_mul;
    if constexpr (EPI == 2) {
        static_assert(KS == 1 && NW == 2 && NTW == 4 && !UP && !N7, "the apply epilogue is a mode of the 256-column Linear tiles");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is past its last LDS fragment read
        la_apply_epilogue(pe, acc, tp.pix0 + 64 * wm, cout_w, lane, reinterpret_cast<float*>(smem) + wave * (32 * LDW), act_scale);
        return;
    }
    if constexpr (EPI == 1) {
        static_assert(KS == 1 && NW == 2 && NTW == 4 && !UP && !N7, "the K'^T V epilogue is a mode of the 256-column Linear tiles");
        la_kv_epilogue(pe, acc, tp.pix0 + 64 * wm, cout_w, lane, act_scale);      // every column block holds k | v
        return;
    }
    if (NTW == 4 && ((p.Cout | p.Csub) & 3) == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is past its last LDS fragment read
        // N7: channels 96..127 of the workgroup (the shared tile) are stored by wn = 0 for mt = 0 and by wn = 1 for mt = 1
        if constexpr (N7)
            conv_epilogue_wide<KS, NW, UP>(pe, acc, tp, cout_w, wm, wave, lane, reinterpret_cast<float*>(smem) + wave * (32 * 128),
                                           reinterpret_cast<float*>(smem) + MW * NW * (32 * 128), wn == 0 ? 1 : 0,
                                           wn == 0 ? 24 : 0, wn == 0 ? 32 : 8);
        else
            conv_epilogue_wide<KS, NW, UP>(pe, acc, tp, cout_w, wm, wave, lane, reinterpret_cast<float*>(smem) + wave * (32 * 128),
                                           reinterpret_cast<float*>(smem) + MW * NW * (32 * 128));
    } else {
        conv_epilogue<KS, NTW>(pe, acc, tp, cout_w, wm, l31, h);
    }
#ifdef FAR_K9_TIMING
    FAR_K9_STAMP(3);                                   // epilogue instructions issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FAR_K9_STAMP(4);                                   // stores acknowledged
    if (threadIdx.x == 0 && blockIdx.x < 65536) g_k9_stamps[12 * blockIdx.x + 9] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int KS, int MW, int NW, int NTW, bool SPLIT, int ST, int UP = 0, bool N7 = false, int EPI = 0>
__global__ __launch_bounds__(64 * MW * NW, (ST == 1 ? 2 : 1)) void k_conv(const ConvArgs p) {
    if constexpr (N7) {                       // two copies of the body, one per wave column: which tile a wave skips is static
        if (((threadIdx.x >> 6) % NW) == 1) conv_body<KS, MW, NW, NTW, SPLIT, ST, UP, N7, 1>(p);
        else conv_body<KS, MW, NW, NTW, SPLIT, ST, UP, N7, 0>(p);
    } else {
        conv_body<KS, MW, NW, NTW, SPLIT, ST, UP, N7, 0, EPI>(p);
    }
}

// max |w| as its bit pattern (non-negative floats order like unsigned integers): one atomicMax per workgroup
__global__ void k_absmax_bits(const float* __restrict__ w, long n, unsigned* __restrict__ out) {
    __shared__ unsigned part[256];
    unsigned m = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned b = __float_as_uint(fabsf(w[i]));
        m = b > m ? b : m;
    }
    part[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d && part[threadIdx.x + d] > part[threadIdx.x]) part[threadIdx.x] = part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(out, part[0]);
}
template <int MODE>      // 0: weight pack scales { 2^w_exp, 2^-(w_exp + 4) }; 1: gradient scales { 2^e, 2^(4 - e) }
__device__ __forceinline__ void write_scales(float amax, float* __restrict__ s) {
    int e = 0;
    if (amax > 0.f && amax <= FLT_MAX) (void)frexpf(amax, &e);
    if (MODE == 0) {
        int w_exp = 14 - e;
        w_exp = w_exp < -60 ? -60 : (w_exp > 60 ? 60 : w_exp);
        s[0] = ldexpf(1.0f, w_exp);
        s[1] = ldexpf(1.0f, -(w_exp + ACT_EXP_DEFAULT));
    } else {
        int ex = 10 - e;
        ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
        s[0] = ldexpf(1.0f, ex);
        s[1] = ldexpf(1.0f, ACT_EXP_DEFAULT - ex);
    }
}
constexpr long AMAX_SINGLE_MAX = 1L << 16;      // elements one workgroup scans (no atomics)

// max |w| of a small tensor by ONE workgroup (no memset, no atomics: one launch), then `finish` turns it into the two scales
template <int MODE>      // 0: weight pack scales { 2^w_exp, 2^-(w_exp + 4) }; 1: gradient scales { 2^e, 2^(4 - e) }
__global__ __launch_bounds__(1024) void k_amax_scale_single(const float* __restrict__ w, long n, float* __restrict__ s) {
    __shared__ unsigned part[1024];
    unsigned m = 0;
    const long n4 = n >> 2;
    const float4* w4 = reinterpret_cast<const float4*>(w);
    for (long i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = w4[i];
        const unsigned a = __float_as_uint(fabsf(v.x)), b = __float_as_uint(fabsf(v.y)), c = __float_as_uint(fabsf(v.z)),
                       d = __float_as_uint(fabsf(v.w));
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
        m = q > m ? q : m;
    }
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 1024) {
        const unsigned b = __float_as_uint(fabsf(w[i]));
        m = b > m ? b : m;
    }
    part[threadIdx.x] = m;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d && part[threadIdx.x + d] > part[threadIdx.x]) part[threadIdx.x] = part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) write_scales<MODE>(__uint_as_float(part[0]), s);
}

// The same for a tensor too large for one workgroup, still ONE launch: every workgroup folds its maximum into a scratch slot
// and takes a ticket; the last one to arrive turns the slot into the scales and leaves it zeroed for its next user.  Slots are
// handed out round-robin on the host, so launches in flight on different streams never share one (1024 slots).
struct AmaxSlot {
    unsigned max_bits, ticket;
};
__device__ AmaxSlot g_amax_slots[1024];

template <int MODE>
__global__ __launch_bounds__(256) void k_amax_scale_multi(const float* __restrict__ w, long n, float* __restrict__ s, int slot) {
    __shared__ unsigned part[256];
    unsigned m = 0;
    const long n4 = n >> 2;
    const float4* w4 = reinterpret_cast<const float4*>(w);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = w4[i];
        const unsigned a = __float_as_uint(fabsf(v.x)), b = __float_as_uint(fabsf(v.y)), c = __float_as_uint(fabsf(v.z)),
                       d = __float_as_uint(fabsf(v.w));
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
        m = q > m ? q : m;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            const unsigned b = __float_as_uint(fabsf(w[i]));
            m = b > m ? b : m;
        }
    part[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d && part[threadIdx.x + d] > part[threadIdx.x]) part[threadIdx.x] = part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        AmaxSlot* const sl = g_amax_slots + slot;
        atomicMax(&sl->max_bits, part[0]);
        __threadfence();
        if (atomicAdd(&sl->ticket, 1u) == gridDim.x - 1) {
            __threadfence();
            const unsigned bits = atomicExch(&sl->max_bits, 0u);
            atomicExch(&sl->ticket, 0u);
            write_scales<MODE>(__uint_as_float(bits), s);
        }
    }
}

std::atomic<unsigned> g_amax_next_slot{0};       // one counter for both modes: a slot has one user at a time

// One launch either way; float4 reads need a 16-byte aligned tensor (anything else takes the three-launch path of the callers).
template <int MODE>
void launch_amax_scale(const float* w, long n, float* out2, hipStream_t stream) {
    if (n <= AMAX_SINGLE_MAX) {
        hipLaunchKernelGGL(k_amax_scale_single<MODE>, dim3(1), dim3(1024), 0, stream, w, n, out2);
        return;
    }
    const int slot = (int)(g_amax_next_slot.fetch_add(1, std::memory_order_relaxed) & 1023u);
    long blocks = (n / 4 + 255) / 256 / 4;                         // >= 4 float4 per thread
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(k_amax_scale_multi<MODE>, dim3((unsigned)blocks), dim3(256), 0, stream, w, n, out2, slot);
}

// s[0] (max |w|, as written above) -> s[0] = 2^w_exp, s[1] = 2^-(w_exp + 4) with w_exp = 14 - exponent(max |w|)  (frexp convention)
__global__ void k_pack_scale(float* __restrict__ s) {
    const float amax = s[0];
    int e = 0;
    if (amax > 0.f && amax <= FLT_MAX) (void)frexpf(amax, &e);
    int w_exp = 14 - e;
    w_exp = w_exp < -60 ? -60 : (w_exp > 60 ? 60 : w_exp);
    s[0] = ldexpf(1.0f, w_exp);
    s[1] = ldexpf(1.0f, -(w_exp + ACT_EXP_DEFAULT));
}

// Packs torch-layout weights [Cout][Cin][KS][KS] (or [Cout][Cin] for linear) into the LDS image the kernel DMAs, in
// execution order: [chunk][tap][k-step][cout block][plane][NT rows][2 slots of 8 channels, slot ^= (row >> 3) & 1]
// fp16, scaled by 2^w_exp (plain operands: [chunk][tap][cout block][k-step][NT rows][...], one slab per tap).
// (w is read through element strides: s_co, s_ci per channel, s_tap per tap of the execution order -- a contiguous torch weight
//  has (Cin * taps, taps, 1); the dgrad image of a convolution reads the SAME tensor with the channel strides swapped and the
//  taps reversed, w_base pointing at the last tap: no flipped / transposed copy is ever made)
__global__ void k_conv_pack(const float* __restrict__ w, long s_co, long s_ci, long s_tap, int Cin, int Cout, int taps, int nchunks,
                            int nblkY, int NT, int planes, float wmul, const float* __restrict__ wmul_dev,
                            _Float16* __restrict__ out, const float* __restrict__ base_scale = nullptr,
                            float* __restrict__ scale_vec = nullptr) {
    if (wmul_dev) wmul = *wmul_dev;                      // the scale chosen on the device (far_conv_pack_auto_f32)
    // the epilogue's scale vector of this image, written here too: base_scale[co] (1 without) x 2^-(w_exp + 4) -- a training step
    // re-packs every weight, and two more launches per weight (ones, multiply) are what this saves
    if (scale_vec && blockIdx.x == 0)
        for (int co = threadIdx.x; co < Cout; co += blockDim.x) scale_vec[co] = (base_scale ? base_scale[co] : 1.0f) * wmul_dev[1];
    const long total = (long)taps * nchunks * 2 * nblkY * NT * 2;
    if (blockIdx.x == 0 && threadIdx.x < 16) out[(size_t)total * 8 * planes + threadIdx.x] = (_Float16)0.f;   // the zero row
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int s = (int)(t & 1); t >>= 1;
        const int nn = (int)(t % NT); t /= NT;
        const int by = (int)(t % nblkY); t /= nblkY;
        const int ks = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % taps);
        const int chunk = (int)(t / taps);
        const int n = by * NT + nn;
        // split: slab = (chunk, tap, k-step), [hi | lo] planes of NT rows; plain: slab = (chunk, tap), [k-step 0 | k-step 1]
        const size_t row = planes == 2 ? ((((size_t)(chunk * taps + tap) * 2 + ks) * nblkY + by) * 2) * NT + nn
                                       : ((((size_t)(chunk * taps + tap) * nblkY + by) * 2 + ks)) * NT + nn;
        _Float16* dst = out + row * 16 + ((s ^ ((nn >> 3) & 1)) * 8);
        for (int e = 0; e < 8; ++e) {
            const int ch = 32 * chunk + 16 * ks + 8 * s + e;
            float v = 0.f;
            if (n < Cout && ch < Cin) v = w[(long)n * s_co + (long)ch * s_ci + (long)tap * s_tap] * wmul;
            const _Float16 hh = (_Float16)v;
            dst[e] = hh;
            if (planes == 2) dst[(size_t)NT * 16 + e] = (_Float16)(v - (float)hh);
        }
    }
}

// ---- the weight images of a whole model, re-packed after an optimizer step in TWO launches (far_pack_table_run): a table row
// per image; k_pack_table_scales reduces max|w| of every row that owns its scale (ticketed like k_amax_scale_multi, a slot
// per row), k_pack_table_images is k_conv_pack over blockIdx.y = row.
struct PackRow {
    const float* w;            // element of tap 0 in execution order
    long s_co, s_ci, s_tap;
    int Cin, Cout, taps, nchunks, nblkY, NT, planes, owns_scale;
    const float* w_all;        // the whole weight tensor (scale owners)
    long n_all;
    float* pack_scale;         // { 2^w_exp, 2^-(w_exp + 4) }
    _Float16* out;
    const float* base_scale;   // or null
    float* scale_vec;          // or null
};
__device__ AmaxSlot g_pack_slots[4096];

__global__ __launch_bounds__(256) void k_pack_table_scales(const PackRow* __restrict__ rows) {
    const PackRow r = rows[blockIdx.y];
    if (!r.owns_scale) return;
    __shared__ unsigned part[256];
    unsigned m = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < r.n_all; i += (long)gridDim.x * 256) {
        const unsigned b = __float_as_uint(fabsf(r.w_all[i]));
        m = b > m ? b : m;
    }
    part[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d && part[threadIdx.x + d] > part[threadIdx.x]) part[threadIdx.x] = part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        AmaxSlot* const sl = g_pack_slots + (blockIdx.y & 4095);
        atomicMax(&sl->max_bits, part[0]);
        __threadfence();
        if (atomicAdd(&sl->ticket, 1u) == gridDim.x - 1) {
            __threadfence();
            const unsigned bits = atomicExch(&sl->max_bits, 0u);
            atomicExch(&sl->ticket, 0u);
            write_scales<0>(__uint_as_float(bits), r.pack_scale);
        }
    }
}

__global__ __launch_bounds__(256) void k_pack_table_images(const PackRow* __restrict__ rows) {
    const PackRow r = rows[blockIdx.y];
    const float wmul = r.pack_scale[0];
    if (r.scale_vec && blockIdx.x == 0)
        for (int co = threadIdx.x; co < r.Cout; co += 256) r.scale_vec[co] = (r.base_scale ? r.base_scale[co] : 1.0f) * r.pack_scale[1];
    const long total = (long)r.taps * r.nchunks * 2 * r.nblkY * r.NT * 2;
    if (blockIdx.x == 0 && threadIdx.x < 16) r.out[(size_t)total * 8 * r.planes + threadIdx.x] = (_Float16)0.f;   // the zero row
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long t = i;
        const int s = (int)(t & 1); t >>= 1;
        const int nn = (int)(t % r.NT); t /= r.NT;
        const int by = (int)(t % r.nblkY); t /= r.nblkY;
        const int ks = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % r.taps);
        const int chunk = (int)(t / r.taps);
        const int n = by * r.NT + nn;
        const size_t row = r.planes == 2 ? ((((size_t)(chunk * r.taps + tap) * 2 + ks) * r.nblkY + by) * 2) * r.NT + nn
                                         : ((((size_t)(chunk * r.taps + tap) * r.nblkY + by) * 2 + ks)) * r.NT + nn;
        _Float16* dst = r.out + row * 16 + ((s ^ ((nn >> 3) & 1)) * 8);
        for (int e = 0; e < 8; ++e) {
            const int ch = 32 * chunk + 16 * ks + 8 * s + e;
            float v = 0.f;
            if (n < r.Cout && ch < r.Cin) v = r.w[(long)n * r.s_co + (long)ch * r.s_ci + (long)tap * r.s_tap] * wmul;
            const _Float16 hh = (_Float16)v;
            dst[e] = hh;
            if (r.planes == 2) dst[(size_t)r.NT * 16 + e] = (_Float16)(v - (float)hh);
        }
    }
}

// Tile configuration by output width: up to 128 channels -> 256 pixels x 128 channels (4 x 1 waves), wider ->
// 128 pixels x 256 channels per block (2 x 2 waves); every wave owns 64 pixels x 128 channels.  (Measured on
// MI355X: 8-wave workgroups of the same wave tile and 64-channel wave tiles were equal or slower.)
struct TileCfg { int mw, nw, nt; };
inline TileCfg cfg_for(int Cout, int stride) {
    // the narrower tile when it pads less (Cout <= 128, and e.g. 384 = 3 x 128: the fused q | k | v of d_model 128)
    const int pad128 = (Cout + 127) / 128 * 128, pad256 = (Cout + 255) / 256 * 256;
    return (stride == 1 && pad128 < pad256) ? TileCfg{4, 1, 128} : TileCfg{2, 2, 256};
}

template <int KS, int MW, int NW, int NTW, bool SPLIT, int ST = 1, int UP = 0, bool N7 = false, int EPI = 0>
int launch_conv(const ConvArgs& a, dim3 grid, hipStream_t stream) {
    using G = Geo<KS, MW, ST>;
    constexpr int PLANES = SPLIT ? 2 : 1;
    constexpr int smem_loop = PLANES * G::A_PLANE + 3 * 2 * 32 * NTW * NW * 32;
    constexpr int smem_epi = EPI == 2 ? MW * NW * 32 * LDW * 4        // la_apply_epilogue: padded rows
                                      : MW * NW * (32 * 128 * 4 + 32 * 4);  // conv_epilogue_wide: 16 KiB per wave + LayerNorm exchange
    constexpr int smem_need = smem_loop + (EPI != 0 ? 64 * MW * 4 : 0);            // gather / padded modes: + the row table behind the loop buffers
    constexpr int smem = smem_need > smem_epi ? smem_need : smem_epi;
    bool cfg_failed = false;
    FAR_ONCE_PER_DEVICE(cfg_failed = hipFuncSetAttribute((const void*)k_conv<KS, MW, NW, NTW, SPLIT, ST, UP, N7, EPI>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess);
    if (cfg_failed) return far_check_launch();
    hipLaunchKernelGGL((k_conv<KS, MW, NW, NTW, SPLIT, ST, UP, N7, EPI>), grid, dim3(64 * MW * NW), smem, stream, a);
    return far_check_launch();
}

template <int KS, bool SPLIT>
int launch_cfg(const TileCfg& c, const ConvArgs& a, dim3 grid, hipStream_t stream, bool small = false) {
    if (KS == 1 && a.up) {                          // the FPN merge variant (fused 2x-upsample residual)
        // rows of whole 32-pixel tiles (the backbone's 320 / 160-pixel rows): the row-walking form of the merge epilogue (UP = 2)
        const bool rowwise = (a.Wo & 31) == 0 && far_get_tuning(13) == 0;
        if (c.mw == 4) return rowwise ? launch_conv<1, 4, 1, 4, SPLIT, 1, 2>(a, grid, stream) : launch_conv<1, 4, 1, 4, SPLIT, 1, 1>(a, grid, stream);
        return rowwise ? launch_conv<1, 2, 2, 4, SPLIT, 1, 2>(a, grid, stream) : launch_conv<1, 2, 2, 4, SPLIT, 1, 1>(a, grid, stream);
    }
    if constexpr (KS == 1) {                        // few rows (a Linear layer of one pair): half-height tiles, twice the workgroups
        if (c.mw == 2 && small) return launch_conv<1, 1, 2, 4, SPLIT>(a, grid, stream);
        if (c.mw == 4 && small) return launch_conv<1, 2, 1, 4, SPLIT>(a, grid, stream);
    }
    if (c.mw == 4) return launch_conv<KS, 4, 1, 4, SPLIT>(a, grid, stream);
    // seven-tile mode: 3x3, one channel block, 193..224 output channels, the 16-byte epilogue (the 196-channel layers)
    if constexpr (KS == 3) {
        if (a.nblkY == 1 && a.Cout > 192 && a.Cout <= 224 && ((a.Cout | a.Csub) & 3) == 0 && far_get_tuning(4) == 0)
            return launch_conv<KS, 2, 2, 4, SPLIT, 1, false, true>(a, grid, stream);
    }
    return launch_conv<KS, 2, 2, 4, SPLIT>(a, grid, stream);
}

// 3x3 stride 2: the (2*8+1) x (2*16+1) input patch takes 94 KB of LDS -> one workgroup per CU (and up to 512 registers)
template <bool SPLIT>
int launch_stride2(const ConvArgs& a, dim3 grid, hipStream_t stream) { return launch_conv<3, 2, 2, 4, SPLIT, 2>(a, grid, stream); }

}  // namespace

extern "C" {

int far_weight_scale_f32(const float* w, long n, float* scale_out, hipStream_t stream);
int far_conv_pack_view_f32(const float* w, long s_co, long s_ci, long s_tap, int Cin, int Cout, int ksize, int stride, int split,
                           const float* scale_in, void* packed, hipStream_t stream);
int far_conv_pack_view_scaled_f32(const float* w, long s_co, long s_ci, long s_tap, int Cin, int Cout, int ksize, int stride, int split,
                                  const float* scale_in, void* packed, const float* base_scale, float* scale_vec_out,
                                  hipStream_t stream);

// Bytes of the packed weight image for a [Cout][Cin][ksize][ksize] weight (split = 1: hi + lo planes).
size_t far_conv_packed_bytes(int Cin, int Cout, int ksize, int stride, int split) {
    if (Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3) || stride < 1 || stride > 2 || (stride == 2 && ksize != 3)) return 0;
    const int NT = cfg_for(Cout, stride).nt;
    const size_t nchunks = (Cin + 31) / 32, nblkY = (Cout + NT - 1) / NT;
    return (size_t)ksize * ksize * nchunks * nblkY * (split ? 2 : 1) * NT * 64 + 32;   // + the zero row padding lanes read
}

// w: torch layout [Cout][Cin][ksize][ksize] fp32.  Every weight is multiplied by 2^w_exp before the fp16 split
// (choose w_exp so that max|w| 2^w_exp is in [2^13, 2^15): the lo parts stay fp16-normal); the caller folds
// 2^-(w_exp + 4) into the `scale` vector passed to far_conv_nhwc_f32.
int far_conv_pack_f32(const float* w, int Cin, int Cout, int ksize, int stride, int w_exp, int split, void* packed,
                      hipStream_t stream) {
    far_clear_errors();
    if (!w || !packed || far_conv_packed_bytes(Cin, Cout, ksize, stride, split) == 0 || w_exp < -60 || w_exp > 60) return FAR_EINVAL;
    const int NT = cfg_for(Cout, stride).nt;
    const int nchunks = (Cin + 31) / 32, nblkY = (Cout + NT - 1) / NT;
    hipLaunchKernelGGL(k_conv_pack, dim3(512), dim3(256), 0, stream, w, (long)Cin * ksize * ksize, (long)ksize * ksize, 1L, Cin, Cout,
                       ksize * ksize, nchunks, nblkY, NT, split ? 2 : 1, ldexpf(1.0f, w_exp), (const float*)nullptr, (_Float16*)packed);
    return far_check_launch();
}

// scale_out (2 device floats) = { 2^w_exp, 2^-(w_exp + 4) } with 2^13 <= max|w| 2^w_exp < 2^14 over the n weights at w.
int far_weight_scale_f32(const float* w, long n, float* scale_out, hipStream_t stream) {
    far_clear_errors();
    if (!w || !scale_out || n <= 0) return FAR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        launch_amax_scale<0>(w, n, scale_out, stream);
    } else {
        hipMemsetAsync(scale_out, 0, 2 * sizeof(float), stream);
        hipLaunchKernelGGL(k_absmax_bits, dim3(256), dim3(256), 0, stream, w, n, reinterpret_cast<unsigned*>(scale_out));
        hipLaunchKernelGGL(k_pack_scale, dim3(1), dim3(1), 0, stream, scale_out);
    }
    return far_check_launch();
}

// Packs a [Cout][Cin][k][k] weight read through element strides (s_co, s_ci, s_tap; w points at the element of tap 0 in
// execution order), multiplied by the device scale scale_in[0] (far_weight_scale_f32).  The dgrad image of a 'same' stride-1
// convolution is the view (s_co, s_ci, s_tap) = (taps, Cin_fwd taps, -1) at w + taps - 1 of the forward weight -- channels
// swapped, taps reversed -- with Cin / Cout exchanged; a transposed Linear weight is (1, K, 0).  No copy of w is made.
int far_conv_pack_view_f32(const float* w, long s_co, long s_ci, long s_tap, int Cin, int Cout, int ksize, int stride, int split,
                           const float* scale_in, void* packed, hipStream_t stream) {
    return far_conv_pack_view_scaled_f32(w, s_co, s_ci, s_tap, Cin, Cout, ksize, stride, split, scale_in, packed, nullptr, nullptr, stream);
}

// The same, also writing the launch's epilogue scale vector: scale_vec_out[co] = base_scale[co] (1 when NULL) * scale_in[1], the
// vector far_conv_nhwc_f32 takes as `scale` (scale_vec_out may be NULL: then exactly far_conv_pack_view_f32).
int far_conv_pack_view_scaled_f32(const float* w, long s_co, long s_ci, long s_tap, int Cin, int Cout, int ksize, int stride, int split,
                                  const float* scale_in, void* packed, const float* base_scale, float* scale_vec_out,
                                  hipStream_t stream) {
    far_clear_errors();
    if (!w || !packed || !scale_in || far_conv_packed_bytes(Cin, Cout, ksize, stride, split) == 0) return FAR_EINVAL;
    const int NT = cfg_for(Cout, stride).nt;
    const int nchunks = (Cin + 31) / 32, nblkY = (Cout + NT - 1) / NT;
    const long items = (long)ksize * ksize * nchunks * 2 * nblkY * NT * 2;
    long blocks = (items + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(k_conv_pack, dim3((unsigned)blocks), dim3(256), 0, stream, w, s_co, s_ci, s_tap, Cin, Cout, ksize * ksize, nchunks, nblkY, NT,
                       split ? 2 : 1, 1.0f, scale_in, (_Float16*)packed, base_scale, scale_vec_out);
    return far_check_launch();
}

// The same with the exponent chosen ON THE DEVICE from max|w| (no host read of the weights: a training step re-packs every
// layer after every optimizer update): scale_out[0] = 2^w_exp (the multiplier applied), scale_out[1] = 2^-(w_exp + 4) (the factor
// the caller folds into far_conv_nhwc_f32's `scale` vector), with 2^13 <= max|w| 2^w_exp < 2^14 (w_exp = 0 for an all-zero weight).
int far_conv_pack_auto_f32(const float* w, int Cin, int Cout, int ksize, int stride, int split, void* packed, float* scale_out,
                           hipStream_t stream) {
    far_clear_errors();
    if (!w || !packed || !scale_out || far_conv_packed_bytes(Cin, Cout, ksize, stride, split) == 0) return FAR_EINVAL;
    const long n = (long)Cout * Cin * ksize * ksize;
    const int rc = far_weight_scale_f32(w, n, scale_out, stream);
    if (rc != FAR_OK) return rc;
    return far_conv_pack_view_f32(w, (long)Cin * ksize * ksize, (long)ksize * ksize, 1, Cin, Cout, ksize, stride, split, scale_out,
                                  packed, stream);
}

// The activation scale of a tensor whose magnitude is not known in advance (gradients: 1e-7 is usual): out = { 2^e, 2^(4 - e) }
// with max|x| 2^e in [2^9, 2^10) -- the top of the split's window, two device floats for far_conv_desc.act_scale_dev.  The
// launch that uses them then needs no pre-scaling pass over its input and no un-scaling pass over its output.
__global__ void k_grad_scale(float* __restrict__ s) {
    const float amax = s[0];
    int e = 0;
    if (amax > 0.f && amax <= FLT_MAX) (void)frexpf(amax, &e);
    int ex = 10 - e;
    ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
    s[0] = ldexpf(1.0f, ex);
    s[1] = ldexpf(1.0f, ACT_EXP_DEFAULT - ex);
}
int far_grad_scale_f32(const float* x, long n, float* out2, hipStream_t stream) {
    far_clear_errors();
    if (!x || !out2 || n <= 0) return FAR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        launch_amax_scale<1>(x, n, out2, stream);
    } else {
        hipMemsetAsync(out2, 0, 2 * sizeof(float), stream);
        hipLaunchKernelGGL(k_absmax_bits, dim3(256), dim3(256), 0, stream, x, n, reinterpret_cast<unsigned*>(out2));
        hipLaunchKernelGGL(k_grad_scale, dim3(1), dim3(1), 0, stream, out2);
    }
    return far_check_launch();
}

// y[n][oy][ox][co] = act(scale[co] * sum_{ky,kx,ci} X[n][s oy+ky-p][s ox+kx-p][ci] * W[co][ci][ky][kx] + shift[co] + res)
// X = x [N][H][W][Cin1] (+ x2 [N][H][W][Cin - Cin1]: the input is their channel concatenation, never materialised;
// x2 = NULL and Cin1 = Cin for a single input); res / y [N][Ho][Wo][Cout] with Ho = (H - 1) / s + 1 (zero padding
// p = ksize / 2), all fp32 NHWC contiguous; stride s = 1, or 2 for ksize 3 (and for ksize 1: x[:, ::2, ::2] read in place, `packed`
// the stride-1 image); Cin % 4 == 0, Cin1 % 8 == 0.
// `scale` must include 2^-(w_exp + 4).  act_exp: the activations are multiplied by 2^act_exp before the fp16 split
// (4 = the default; the kernel corrects `scale` by 2^(4 - act_exp)): inputs up to 65504 / 2^act_exp survive the split, and
// act_exp can be lowered (down to -24) for tensors beyond 4094 at the price of the absolute resolution of tiny values
// (2^-24 2^-act_exp).  overflow: device int, OR-ed with 1 when the launch produced a non-finite accumulator (an input beyond
// that range), NULL to skip the test.  act: 0 none, 1 ReLU, 2 LeakyReLU(slope).  A linear layer is ksize = 1 with
// N = H = 1, W = rows.  out_planes > 1: output channel co goes to plane co / (Cout / out_planes) of y, laid out
// [out_planes][N][Ho][Wo][Cout / out_planes] (fused projections, e.g. q | k | v); res, if given, has y's layout, or
// with res_group = G > 1 (linear layers) is [rows / G][Cout]: every group of G consecutive rows shares one residual
// row (fine_preprocess.py:52-57: the coarse feature of a match, repeated over its 25 window tokens).
// y must alias none of the inputs.
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

}  // extern "C"

namespace {
struct KvMode {                                       // far_linear_kv_f16s (epi 1) / far_linear_q_apply_f16s (epi 2)
    int epi;
    float* part;
    int S, valid;                                     // rows per image in the launch's geometry (a multiple of 64) / real tokens per image
    const unsigned char* img;
    float eps, vlen;
    const long *g_b = nullptr, *g_cell = nullptr;     // epi 3 (far_linear_gather_f16s)
    int g_wc = 0, g_W = 0, g_stride = 0, g_Hf = 0, g_Wf = 0;
};

int conv_nhwc_impl(const far_conv_desc* desc, const KvMode* kvm, hipStream_t stream) {
    if (!desc) return FAR_EINVAL;
    const far_conv_desc& d = *desc;
    const float *x = d.x, *x2 = d.x2, *scale = d.scale, *shift = d.shift, *res = d.res;
    const float *ln_gamma = d.ln_gamma, *ln_beta = d.ln_beta, *post_res = d.post_res, *up = d.up;
    const void* packed = d.packed;
    float* y = d.y;
    const long N = d.N;
    const int H = d.H, W = d.W, Cin = d.Cin, Cin1 = d.Cin1, Cout = d.Cout, ksize = d.ksize, stride = d.stride;
    const int act = d.act, split = d.split, out_planes = d.out_planes, res_group = d.res_group;
    const float slope = d.slope, ln_eps = d.ln_eps;
    far_clear_errors();
    if (d.act_exp < -24 || d.act_exp > 8) return FAR_EINVAL;
    if (N == 0) return FAR_OK;
    // ksize 1 with stride 2: the 1x1 convolution of x[:, ::2, ::2] (the down-sampling shortcut) read in place; the weight image is
    // the stride-1 one
    const bool sub2 = ksize == 1 && stride == 2;
    const int pstride = sub2 ? 1 : stride;
    if (!x || !packed || !scale || (!y && !(kvm && kvm->epi == 1)) || N < 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 ||
        (ksize != 1 && ksize != 3) || stride < 1 || stride > 2 || act < 0 || act > 2 ||
        (sub2 && (kvm || up || x2 || res_group != 1 || N * (long)((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) > 0x7fffffffL)) ||
        (ksize == 3 && N * (long)H * W > 0x7fffffffL) ||          // 3x3: input pixel indices are 32-bit (stage_pixels)
        (act == 2 && !(slope >= 0.f && slope <= 1.f)) || x == y || (x2 && x2 == y) || out_planes < 1 || Cout % out_planes)
        return FAR_EINVAL;
    if (res_group < 1 || (res_group > 1 && (!res || ksize != 1 || out_planes != 1 || (N * H * W) % res_group))) return FAR_EINVAL;
    if (x2 ? (Cin1 <= 0 || Cin1 >= Cin || (Cin1 & 7)) : (Cin1 != Cin)) return FAR_EINVAL;
    // fused 2x-upsample residual: 1x1 convolutions only, even H x W >= 2 x 32 (a 32-pixel tile row wraps at most once),
    // one output tensor, no other residual / LayerNorm, 16-byte channel groups, a tensor the 32-bit row offsets can span
    if (up && (ksize != 1 || (H & 1) || (W & 1) || W < 32 || out_planes != 1 || res || ln_gamma || post_res || (Cout & 3) ||
               up == y || (long)(H / 2) * (W / 2) * Cout > 0x7fffffffL))
        return FAR_EINVAL;
    ConvArgs a;
    a.x = x; a.x2 = x2; a.Cin1 = Cin1; a.w = (const unsigned char*)packed; a.scale = scale;
    a.zeros = reinterpret_cast<const float*>((const unsigned char*)packed + far_conv_packed_bytes(Cin, Cout, ksize, pstride, split) - 32); a.shift = shift; a.res = res; a.res_group = res_group; a.y = y;
    a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.ln_eps = ln_eps; a.post_res = post_res; a.up = up;
    a.H = H; a.W = W; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1; a.Cin = Cin; a.Cout = Cout; a.Csub = Cout / out_planes;
    a.npix = N * a.Ho * a.Wo;
    if (kvm && kvm->epi != 3 && kvm->valid != kvm->S) a.npix = a.npix / kvm->valid * kvm->S;      // padded geometry: images x rounded-up length
    const TileCfg c = cfg_for(Cout, pstride);
    if ((ln_gamma || post_res) && (!ln_gamma || !ln_beta || Cout != c.nt || out_planes != 1 || (Cout & 3) || post_res == y))
        return FAR_EINVAL;                    // the fused LayerNorm needs the whole channel row in one block (Cout 128 or 256)
    const int th = 4 * c.mw;
    a.tilesX = (a.Wo + TW - 1) / TW; a.tilesY = (a.Ho + th - 1) / th;
    a.nchunks = (Cin + 31) / 32;
    a.nblkY = (Cout + c.nt - 1) / c.nt;
    a.act = act; a.slope = slope;
    a.act_scale = ldexpf(1.0f, d.act_exp); a.out_mul = ldexpf(1.0f, ACT_EXP_DEFAULT - d.act_exp); a.overflow = d.overflow;
    a.scale_dev = d.act_scale_dev;
    a.kv_part = kvm ? kvm->part : nullptr; a.kv_S = kvm ? kvm->S : 0; a.kv_valid = kvm ? kvm->valid : 0;
    a.sub2 = sub2 ? 1 : 0;
    a.g_b = a.g_cell = nullptr; a.g_wc = a.g_W = a.g_stride = a.g_Hf = a.g_Wf = 0;
    a.kv_img = kvm ? kvm->img : nullptr; a.la_eps = kvm ? kvm->eps : 0.f; a.la_vlen = kvm ? kvm->vlen : 0.f;
    long nbx = ksize == 1 ? (a.npix + 64 * c.mw - 1) / (64 * c.mw) : N * a.tilesX * a.tilesY;
    // a Linear layer over the tokens of one or two images: the few-row kernel (linear_small_f16s.hip; same image, same arithmetic)
    if (ksize == 1 && !sub2 && !kvm && !up && !x2 && !ln_gamma && !post_res && res_group == 1 && split && nbx * a.nblkY < 512 &&
        far_get_tuning(7) == 0 && far_linear_small_covers(a.npix, Cin, Cout)) {
        LinSmallArgs s;
        s.x = x; s.w = a.w; s.scale = scale; s.shift = shift; s.res = res; s.y = y; s.rows = a.npix;
        s.Cin = Cin; s.Cout = Cout; s.Csub = a.Csub; s.NT = c.nt; s.nblkY = a.nblkY; s.act = act; s.slope = slope;
        s.act_scale = a.act_scale; s.out_mul = a.out_mul; s.scale_dev = a.scale_dev; s.overflow = a.overflow;
        return far_linear_small_launch(s, stream);
    }
    // a Linear layer over the tokens of one pair fills a fraction of the CUs with full-height tiles: halve them
    const bool small = ksize == 1 && !up && nbx * a.nblkY < 192 && far_get_tuning(7) == 0;
    if (small) nbx = (a.npix + 32 * c.mw - 1) / (32 * c.mw);
    a.ntiles = nbx;
    if (nbx * a.nblkY > 0x7fffffffL) return FAR_EINVAL;
    dim3 grid((unsigned)(nbx * a.nblkY));
    if (stride == 2 && ksize == 3) return split ? launch_stride2<true>(a, grid, stream) : launch_stride2<false>(a, grid, stream);
    if (ksize == 3) return split ? launch_cfg<3, true>(c, a, grid, stream) : launch_cfg<3, false>(c, a, grid, stream);
    if (kvm && kvm->epi == 3) {                    // gather mode: rows read from the fine map through the window indices
        if (!split) return FAR_EINVAL;
        a.g_b = kvm->g_b; a.g_cell = kvm->g_cell; a.g_wc = kvm->g_wc; a.g_W = kvm->g_W; a.g_stride = kvm->g_stride;
        a.g_Hf = kvm->g_Hf; a.g_Wf = kvm->g_Wf;
        if (small) { nbx = (a.npix + 64 * c.mw - 1) / (64 * c.mw); a.ntiles = nbx; grid = dim3((unsigned)(nbx * a.nblkY)); }   // full-height tiles only
        return c.mw == 4 ? launch_conv<1, 4, 1, 4, true, 1, 0, false, 3>(a, grid, stream)
                         : launch_conv<1, 2, 2, 4, true, 1, 0, false, 3>(a, grid, stream);
    }
    if (kvm) {                                     // (validated by far_linear_kv_f16s: 256-column blocks)
        if (c.mw != 2) return FAR_EINVAL;
        if (!split) {                              // plain fp16 operands in the projection's K loop; the epilogue's own small products stay split
            if (kvm->epi == 2)
                return small ? launch_conv<1, 1, 2, 4, false, 1, false, false, 2>(a, grid, stream)
                             : launch_conv<1, 2, 2, 4, false, 1, false, false, 2>(a, grid, stream);
            return small ? launch_conv<1, 1, 2, 4, false, 1, false, false, 1>(a, grid, stream)
                         : launch_conv<1, 2, 2, 4, false, 1, false, false, 1>(a, grid, stream);
        }
        if (kvm->epi == 2)
            return small ? launch_conv<1, 1, 2, 4, true, 1, false, false, 2>(a, grid, stream)
                         : launch_conv<1, 2, 2, 4, true, 1, false, false, 2>(a, grid, stream);
        return small ? launch_conv<1, 1, 2, 4, true, 1, false, false, 1>(a, grid, stream)
                     : launch_conv<1, 2, 2, 4, true, 1, false, false, 1>(a, grid, stream);
    }
    return split ? launch_cfg<1, true>(c, a, grid, stream, small) : launch_cfg<1, false>(c, a, grid, stream, small);
}
}  // namespace

extern "C" {

int far_conv_nhwc_f32(const far_conv_desc* desc, hipStream_t stream) { return conv_nhwc_impl(desc, nullptr, stream); }

// The k | v projections of a LoFTR encoder layer at d_model 256 / 8 heads in ONE launch that never writes k or v: desc describes a
// Linear layer (ksize 1, N = H = 1, W = rows, split operands) with Cout = 512, out_planes = 2, whose packed weight is Wk and Wv
// interleaved head by head (weight rows 64 j + [0, 32) = Wk's rows of head j, 64 j + [32, 64) = Wv's); y is not used (NULL).
// rows = n_img * S tokens, image after image (S >= 64); kv [n_img][256][33] = K'^T (V / S) per head (33rd column: sum of K'), the
// state far_linear_attention_apply_f32 consumes -- what far_linear_attention_f32 computes from k and v in its first two launches.
// kv_img (optional, far_linear_kv_image_bytes(n_img) bytes): the same state as far_linear_q_apply_f16s's operands (x 2^act_exp).
// ws: far_linear_kv_workspace_bytes(rows, S) bytes.  No masks, no residual / LayerNorm / activation on this launch.
size_t far_linear_kv_workspace_bytes(long rows, int S) {
    if (rows <= 0 || S < 64 || rows % S) return 0;
    return (size_t)(rows / S) * ((S + 63) / 64) * 256 * 33 * sizeof(float);
}
size_t far_linear_kv_image_bytes(long n_img) { return n_img > 0 ? (size_t)n_img * KV_IMG_BYTES : 0; }

int far_linear_kv_f16s(const far_conv_desc* desc, int S, void* ws, float* kv, void* kv_img, hipStream_t stream) {
    if (!desc) return FAR_EINVAL;
    const far_conv_desc& d = *desc;
    far_clear_errors();
    const long rows = d.N * d.H * d.W;
    if (rows == 0) return FAR_OK;
    if (!ws || !kv || S < 64 || rows < 0 || rows % S || d.ksize != 1 || d.stride != 1 || d.Cout != 512 ||
        d.out_planes != 2 || d.act != 0 || d.res || d.ln_gamma || d.post_res || d.up || d.x2 || d.res_group != 1 || rows / S > 65535 ||
        d.act_scale_dev)
        return FAR_EINVAL;
    const int Sp = (S + 63) / 64 * 64;                   // the launch's rows per image: whole 64-row blocks (the row table skips the rest)
    if (rows / S * (long)Sp > 0x7fffffffL) return FAR_EINVAL;
    KvMode m{1, (float*)ws, Sp, S, nullptr, 0.f, 0.f};
    const int rc = conv_nhwc_impl(desc, &m, stream);
    if (rc != FAR_OK) return rc;
    const int per = 256 * 33;
    hipLaunchKernelGGL(k_kv_blocks_reduce, dim3((per + 255) / 256, (unsigned)(rows / S)), dim3(256), 0, stream, (const float*)ws, Sp,
                       per, kv, (unsigned char*)kv_img, ldexpf(1.0f, d.act_exp));
    return far_check_launch();
}

// The q projection of the same layer with LinearAttention's second half (linear_attention.py:46-50) in its epilogue: desc describes
// a Linear layer (ksize 1, N = H = 1, W = rows, split operands, Cout = 256, out_planes = 1, no activation / residual / LayerNorm)
// with the plain Wq image; y [rows][256] receives the attention MESSAGE (Q' KV) Z S -- q itself is never stored.  kv_img: the
// operand image far_linear_kv_f16s wrote for the SOURCE tokens under the same act_exp (image i serves rows [i L, (i + 1) L));
// L = tokens per image on the query side (>= 64), S = v_length of the source (linear_attention.py:43, 50).
int far_linear_q_apply_f16s(const far_conv_desc* desc, int L, int S, const void* kv_img, float eps, hipStream_t stream) {
    if (!desc) return FAR_EINVAL;
    const far_conv_desc& d = *desc;
    far_clear_errors();
    const long rows = d.N * d.H * d.W;
    if (rows == 0) return FAR_OK;
    if (!kv_img || !d.y || L < 64 || S <= 0 || rows < 0 || rows % L || d.ksize != 1 || d.stride != 1 ||
        d.Cout != 256 || d.out_planes != 1 || d.act != 0 || d.res || d.ln_gamma || d.post_res || d.up || d.x2 || d.res_group != 1 ||
        d.act_scale_dev)
        return FAR_EINVAL;
    const int Lp = (L + 63) / 64 * 64;
    if (rows / L * (long)Lp > 0x7fffffffL) return FAR_EINVAL;
    KvMode m{2, nullptr, Lp, L, (const unsigned char*)kv_img, eps, (float)S};
    return conv_nhwc_impl(desc, &m, stream);
}

// merge_feat of FinePreprocess (fine_preprocess.py:40-57) without the window tensor: desc describes a Linear layer (ksize 1, N = H = 1,
// W = rows, split operands; residual / res_group / activation as far_conv_nhwc_f32) whose input rows are NOT stored anywhere -- row r
// is token r % (W W) of window r / (W W), i.e. pixel (cy stride - W / 2 + ky, cx stride - W / 2 + kx) of image b_ids[window] of the fine
// map desc.x [n_img][Hf][Wf][Cin] (NHWC, zero outside), (cy, cx) = divmod(cell_ids[window], wc), (ky, kx) = divmod(token, W): what
// far_fine_gather_f32 would have written as [windows][W W][Cin] for this launch to read back.  rows % (W W) == 0.
int far_linear_gather_f16s(const far_conv_desc* desc, const int64_t* b_ids, const int64_t* cell_ids, int wc, int W, int stride,
                           long n_img, int Hf, int Wf, hipStream_t stream) {
    if (!desc) return FAR_EINVAL;
    const far_conv_desc& d = *desc;
    far_clear_errors();
    const long rows = d.N * d.H * d.W;
    if (rows == 0) return FAR_OK;
    if (!b_ids || !cell_ids || wc <= 0 || W <= 0 || W > 15 || stride <= 0 || n_img <= 0 || Hf <= 0 || Wf <= 0 || rows < 0 ||
        rows % (W * W) || rows > 0x7fffffffL || n_img * Hf * (long)Wf > 0x7fffffffL || d.ksize != 1 || d.stride != 1 || !d.split ||
        d.x2 || d.up || d.ln_gamma || d.post_res || d.act_scale_dev || d.out_planes != 1)
        return FAR_EINVAL;
    KvMode m{3, nullptr, 0, 0, nullptr, 0.f, 0.f};
    m.g_b = reinterpret_cast<const long*>(b_ids); m.g_cell = reinterpret_cast<const long*>(cell_ids);
    m.g_wc = wc; m.g_W = W; m.g_stride = stride; m.g_Hf = Hf; m.g_Wf = Wf;
    return conv_nhwc_impl(desc, &m, stream);
}

// ---- every weight image of a model in two launches (training: all weights change at every optimizer step).
// far_pack_item (include/far_hip.h) describes one image as far_conv_pack_view_scaled_f32's arguments do; scale_owner = the
// index of the item whose max|w| reduction this image uses (its own index: it reduces w_all[0 .. n_all) itself; a dgrad /
// transposed image names its forward image and shares that item's pack_scale pointer).
struct far_pack_item {          // mirrors include/far_hip.h
    const float* w;
    long s_co, s_ci, s_tap;
    int Cin, Cout, ksize, stride, split, scale_owner;
    const float* w_all;
    long n_all;
    float* pack_scale;
    void* packed;
    const float* base_scale;
    float* scale_vec;
};
long far_pack_table_bytes(int n) { return n > 0 ? (long)n * (long)sizeof(PackRow) : 0; }

int far_pack_table_build(const far_pack_item* items, int n, void* table_dev, hipStream_t stream) {
    far_clear_errors();
    if (!items || !table_dev || n <= 0 || n > 4096) return FAR_EINVAL;
    std::vector<PackRow> rows((size_t)n);
    for (int i = 0; i < n; ++i) {
        const far_pack_item& it = items[i];
        if (!it.w || !it.packed || !it.pack_scale || it.scale_owner < 0 || it.scale_owner >= n ||
            far_conv_packed_bytes(it.Cin, it.Cout, it.ksize, it.stride, it.split) == 0)
            return FAR_EINVAL;
        const bool owns = it.scale_owner == i;
        if (owns ? (!it.w_all || it.n_all <= 0) : (items[it.scale_owner].pack_scale != it.pack_scale)) return FAR_EINVAL;
        PackRow& r = rows[(size_t)i];
        const int NT = cfg_for(it.Cout, it.stride).nt;
        r.w = it.w; r.s_co = it.s_co; r.s_ci = it.s_ci; r.s_tap = it.s_tap;
        r.Cin = it.Cin; r.Cout = it.Cout; r.taps = it.ksize * it.ksize; r.nchunks = (it.Cin + 31) / 32; r.nblkY = (it.Cout + NT - 1) / NT;
        r.NT = NT; r.planes = it.split ? 2 : 1; r.owns_scale = owns ? 1 : 0;
        r.w_all = it.w_all; r.n_all = it.n_all; r.pack_scale = it.pack_scale; r.out = (_Float16*)it.packed;
        r.base_scale = it.base_scale; r.scale_vec = it.scale_vec;
    }
    // (the copy is from pageable host memory: the runtime stages it before returning, `rows` may die afterwards)
    if (hipMemcpyAsync(table_dev, rows.data(), rows.size() * sizeof(PackRow), hipMemcpyHostToDevice, stream) != hipSuccess)
        return far_check_launch();
    return FAR_OK;
}

int far_pack_table_run(const void* table_dev, int n, hipStream_t stream) {
    far_clear_errors();
    if (!table_dev || n <= 0 || n > 4096) return FAR_EINVAL;
    const PackRow* rows = reinterpret_cast<const PackRow*>(table_dev);
    hipLaunchKernelGGL(k_pack_table_scales, dim3(16, (unsigned)n), dim3(256), 0, stream, rows);
    hipLaunchKernelGGL(k_pack_table_images, dim3(32, (unsigned)n), dim3(256), 0, stream, rows);
    return far_check_launch();
}

#ifdef FAR_K9_TIMING
// s_memtime ticks vs the 100 MHz s_memrealtime clock (is a tick a shader cycle?)
__global__ void k_tick_probe(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1;
    do { t1 = __builtin_amdgcn_s_memtime(); } while (t1 - t0 < ticks);
    out[0] = t1 - t0; out[1] = __builtin_amdgcn_s_memrealtime() - r0;
}
int far_k9_tick_probe(unsigned long long ticks, unsigned long long* host2) {
    unsigned long long* d;
    if (hipMalloc(&d, 16) != hipSuccess) return -5;
    hipLaunchKernelGGL(k_tick_probe, dim3(1), dim3(64), 0, 0, ticks, d);
    hipMemcpy(host2, d, 16, hipMemcpyDeviceToHost);
    hipFree(d);
    return 0;
}
int far_k9_timing_dump(void* host, int nblocks) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k9_stamps), (size_t)nblocks * 12 * sizeof(unsigned long long)) == hipSuccess ? 0 : -5;
}
#endif

}  // extern "C"
