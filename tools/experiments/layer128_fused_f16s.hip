// STATUS: PARKED EXPERIMENT (round 4) -- not built into libfar_hip.so.  Built, packed and wired once (the Python side is in
// layer128_ops.py, the driver in layer128_try.py): against the two launches it replaces (K14 + K13, 2.31 ms per 61 k windows) it
// measured 3.07 ms in the 256-register build, which spills (692 bytes of scratch per lane = as many bytes again as the windows the
// kernel streams) AND is not run-to-run deterministic beyond a few workgroups (also with one workgroup per CU), and 3.42 ms,
// correct and deterministic at every size, in a 512-register build (one workgroup per CU).  The layouts below are right (2.8e-7 of
// the two-launch result); what is missing is a register allocation that holds the message (64) next to GEMM 1's accumulators
// without spilling into the hand-counted LDS-DMA phases.  DESIGN.md section 7 has the numbers.
// K21: a whole LoFTR encoder layer at d_model = 128 on the 25-token fine windows as ONE kernel: K14's attention block
// (attn_block_f16s.hip: q / k / v projections, linear attention, merge, norm1) followed, in the same wave and on the same 32-row
// tile, by K13's MLP block (mlp_fused_f16s.hip: mlp[0] + ReLU + mlp[2] + norm2 + residual).
//
// Replaces mp3d_loftr/src/loftr/loftr_module/transformer.py:44-67 (LoFTREncoderLayer.forward, both halves).
// As two launches the layer writes the normalised message (781 MB per 61 k windows) for the second launch to read back, and
// the second launch reads x again: three of the layer's six passes over HBM, in kernels that run at the sum of their HBM and
// matrix floors.  Here the message stays in registers:
//   * the merge GEMM runs TRANSPOSED -- the same weight fragments as K14, operands exchanged (the A and B layouts of the
//     32 x 32 x 16 MFMA are mirror images) -- so D[m = channel][n = token]: a lane owns one token and 64 of its 128 channels;
//   * norm1 is then an in-lane sum and one exchange with the lane's twin (lane ^ 32), no DPP chains;
//   * the normalised message in that layout IS the B operand of K13's transposed first GEMM (H^T = W0 [x | msg]^T) for the
//     message half of its K dimension, with W0's message columns packed in the accumulator order 4 h + (e & 3) + 8 (e >> 2) (as
//     K13 packs W2); that half runs first (the message registers die as it proceeds: 64 + 128 accumulators at the peak), then
//     the x half from x chunks requested again by LDS-DMA (L2-warm: the attention block read them a few microseconds earlier);
//   * GEMM 2, norm2, + x and the store are K13's.
// One weight image: K14's 16 slabs, then per half of the hidden dimension 4 message + 4 x + 4 GEMM-2 slabs, through one 3-slot LDS
// ring (40 phases, one barrier each).
// The MLP half works on K14's window-padded tiles (25 of 32 rows valid) where K13 alone packs rows densely: 28 % more matrix
// work on that half, paid for by the three passes that disappear.
#include "common.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int DM = 128;                  // d_model
constexpr int CT = DM / 32;              // 32-channel tiles (two 16-channel heads each)
constexpr int SLAB = 16384;              // two k-steps x 4 tiles x 2 planes x 1 KiB
constexpr int NSLAB = 40;                // K14's 16: [k c0][v c0] .. [k c3][v c3][q c0..c3][merge t0..t3]; then per hidden half: GEMM 1 message slabs (4), x slabs (4), GEMM 2 (4)
constexpr int RING = 3;
constexpr int WAVES = 4;
constexpr int PIECES = 16 / WAVES;
constexpr float ACT_SCALE = 16.0f;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
// Every phase waits for ALL of this wave's outstanding requests (vmcnt(0)) before the barrier.  The requests are issued two
// slabs / one chunk ahead, so the youngest is a phase old and the wait costs ~2 %.  Counting them instead (vmcnt(4) / (8):
// "slab p and this phase's chunk have landed once only the younger requests remain") worked in K13 but gave stale LDS
// reads here -- whole windows wrong in ~3 % of the windows of every launch beyond the first round of workgroups -- whenever
// six or more requests were allowed to stay in flight across a barrier of the k / v phases; the run-to-run determinism test
// of tests/test_attn_block_gpu.py is what found it.
__device__ __forceinline__ float elu1(float x) { return (x > 0.f ? x : expm1f(x)) + 1.f; }   // F.elu(x) + 1

__device__ __forceinline__ void split_regs(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {                       // packed conversions (common.h: split2)
        f16x2 h, l;
        split2(f32x2{v[i], v[i + 1]}, h, l);
        hi[i] = h.x; hi[i + 1] = h.y;
        lo[i] = l.x; lo[i + 1] = l.y;
    }
}
__device__ __forceinline__ void split8(const float4& u, const float4& v, float scale, f16x8& hi, f16x8& lo) {
    const float x[8] = {u.x * scale, u.y * scale, u.z * scale, u.w * scale, v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    split_regs(x, hi, lo);
}
__device__ __forceinline__ f32x16 mma3(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
}

#define FAR_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float sum32(float v) {
    v += FAR_DPP_F(v, 0xB1);
    v += FAR_DPP_F(v, 0x4E);
    v += FAR_DPP_F(v, 0x141);
    v += FAR_DPP_F(v, 0x140);
    return v + shfl_xor_f(v, 16);
}

struct Scales { float k, v, q, m; };     // accumulator -> value: 2^-(w_exp + 4) per weight tensor

__global__ __launch_bounds__(256, 2) void k_layer128(const float* __restrict__ x, const float* __restrict__ src,
                                                     const unsigned char* __restrict__ wimg, long nwin, int L, int S, Scales sc,
                                                     float attn_eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float ln_eps, float hscale, float oscale, const float* __restrict__ gamma2,
                                                     const float* __restrict__ beta2, float ln2_eps, float* __restrict__ out,
                                                     int* __restrict__ overflow) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[RING * SLAB];
    __shared__ __attribute__((aligned(16))) unsigned char xs[WAVES * 4096];
#ifdef FAR_K21_ONE_WG
    __shared__ unsigned char k21_pad[40 * 1024];
    if (nwin < 0) k21_pad[threadIdx.x] = 1;
#endif
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long win = (long)blockIdx.x * WAVES + wave;
    const bool live = win < nwin;
    if (!live) win = nwin - 1;                                 // a spare wave of the last workgroup: works on a valid window, stores nothing
    const unsigned ring_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)ring);
    const unsigned xs_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)(xs + wave * 4096));
    const unsigned char* wsrc = wimg + (size_t)lane * 16;
    auto request_w = [&](int s) {
        const unsigned dst = ring_base + (unsigned)((s % RING) * SLAB);
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            glds16(wsrc + (size_t)s * SLAB + (wave + WAVES * i) * 1024, dst + (wave + WAVES * i) * 1024);
    };
    // chunk j: 32 channels of this window's rows -> the wave's 4 KiB, row-major [32 rows][8 pieces of 16 B], source-side
    // swizzle (piece q of row r holds source piece q ^ ((r ^ (r >> 3)) & 7)).  j < 4: source channels 32 j; j >= 4: x.
    const int rr = lane >> 3, q8 = lane & 7;
    const long lastrow_s = nwin * S - 1, lastrow_x = nwin * L - 1;
    auto request_x = [&](int j) {
        const bool is_src = j < 4;
        const float* base = (is_src ? src : x) + 32 * (j & 3);
        const long r0 = win * (is_src ? S : L), last = is_src ? lastrow_s : lastrow_x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + rr;
            long gr = r0 + r;
            gr = gr < last ? gr : last;                        // rows past the window are masked below; keep the address valid
            glds16(base + gr * DM + 4 * (q8 ^ ((r ^ (r >> 3)) & 7)), xs_base + i * 1024);
        }
    };
    request_x(0);
    request_w(0);
    request_w(1);
    const unsigned char* xrd = xs + wave * 4096 + (l31 >> 3) * 1024 + (l31 & 7) * 128;
    const int sw = (l31 ^ (l31 >> 3)) & 7;
    f16x8 xh[2], xl[2];                                       // the current chunk of this lane's row: two k-steps of (hi, lo)
    auto read_chunk = [&](int next) {                          // read out the landed chunk, then request chunk `next` (< 8) into the same 4 KiB
        float4 raw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) raw[i] = *reinterpret_cast<const float4*>(xrd + (((4 * h + i) ^ sw) * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (next < 8) request_x(next);
        split8(raw[0], raw[1], ACT_SCALE, xh[0], xl[0]);
        split8(raw[2], raw[3], ACT_SCALE, xh[1], xl[1]);
    };
    // one slab of a projection GEMM: two k-steps x four 32-channel tiles.  TRANSPOSED: D[m = channel][n = token] (weights are
    // the A operand), else D[m = token][n = channel]
    auto gemm_slab = [&](int s, f32x16 (&acc)[CT], auto transposed) {
        constexpr bool TR = decltype(transposed)::value;
        const unsigned char* slab = ring + (s % RING) * SLAB + lane * 16;
        f16x8 wh[3], wl[3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wh[i] = *reinterpret_cast<const f16x8*>(slab + i * 2048);
            wl[i] = *reinterpret_cast<const f16x8*>(slab + i * 2048 + 1024);
        }
#pragma unroll
        for (int i = 0; i < 2 * CT; ++i) {                     // i = k-step * CT + tile
            if (i + 2 < 2 * CT) {
                wh[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + (i + 2) * 2048);
                wl[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + (i + 2) * 2048 + 1024);
            }
            const int ks = i / CT, t = i % CT;
            if (TR) acc[t] = mma3(wh[i % 3], wl[i % 3], xh[ks], xl[ks], acc[t]);
            else acc[t] = mma3(xh[ks], xl[ks], wh[i % 3], wl[i % 3], acc[t]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
    };
    auto begin_phase = [&](int) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");                               // slab p complete and visible; slot (p + 2) % 3 free
    };
    auto zero = [&](f32x16 (&a)[CT]) {
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[t][r] = 0.f;
    };

    // ------------------------------------------------------------------ k, v = source W^T   (phases 0..7)
    f32x16 ka[CT], va[CT];
    zero(ka);
    zero(va);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        begin_phase(2 * c);
        read_chunk(c + 1);                                     // source chunk c; next: source chunk c + 1 (c = 3: x chunk 0)
        request_w(2 * c + 2);
        gemm_slab(2 * c, ka, std::false_type{});
        begin_phase(2 * c + 1);
        request_w(2 * c + 3);
        gemm_slab(2 * c + 1, va, std::false_type{});
    }
    // K' = elu(k) + 1, V = v / S on the rows of the window (registers: token mfma32_row(r, h); lane: channel 32 t + l31)
    float ksum[CT];
    const float fS = (float)S;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        float s_ = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = mfma32_row(r, h) < S;
            const float kk = ok ? elu1(ka[t][r] * sc.k) : 0.f;
            ka[t][r] = kk * ACT_SCALE;
            va[t][r] = ok ? (va[t][r] * sc.v) / fS * ACT_SCALE : 0.f;      // values / v_length (linear_attention.py:43)
            s_ += kk;
        }
        ksum[t] = s_ + shfl_xor_f(s_, 32);                      // lane l31 (either half): sum_s K'_s [channel 32 t + l31]
    }
    // KV[d][e] = sum_s K'[s][d] V[s][e] per 32-channel tile: D[m = d][n = e], contraction over the registers (tokens)
    f32x16 kv[CT];
    zero(kv);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float a8[8], b8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { a8[e] = ka[t][8 * u + e]; b8[e] = va[t][8 * u + e]; }
            f16x8 ah, al, bh, bl;
            split_regs(a8, ah, al);
            split_regs(b8, bh, bl);
            kv[t] = mma3(ah, al, bh, bl, kv[t]);
        }
    // kv[t][r]: d = mfma32_row(r, h), e = l31; drop the cross-head quarters (head = channel / 16) and the two 2^4 scales
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            kv[t][r] = ((r >= 8) == (l31 >= 16)) ? kv[t][r] * (1.0f / ACT_SCALE) : 0.f;      // left scaled by 2^4 for its split

    // ------------------------------------------------------------------ q^T = Wq x^T   (phases 8..11)
    f32x16 qa[CT];
    zero(qa);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        begin_phase(8 + c);
        read_chunk(c + 5);                                     // x chunk c; next: x chunk c + 1 (none after the last)
        request_w(10 + c);
        gemm_slab(8 + c, qa, std::true_type{});
    }
    // Q' = elu(q) + 1 (lane = token l31; registers: channel 32 t + mfma32_row(r, h));  den = Q' . ksum per head
    f32x16 ma[CT];                                             // message^T: D[m = e][n = token]
    zero(ma);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        float den0 = 0.f, den1 = 0.f;                          // heads 2 t and 2 t + 1
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float qq = elu1(qa[t][r] * sc.q);
            const float ks = __shfl(ksum[t], mfma32_row(r, h), 64);          // ksum of this register's channel
            if (r < 8) den0 += qq * ks; else den1 += qq * ks;
            qa[t][r] = qq * ACT_SCALE;
        }
        den0 += shfl_xor_f(den0, 32);
        den1 += shfl_xor_f(den1, 32);
        // message^T[e][token] = sum_d KV[d][e] Q'[token][d]:  A = KV block (lane = e, registers = d), B = Q'^T (lane = token)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float a8[8], b8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { a8[e] = kv[t][8 * u + e]; b8[e] = qa[t][8 * u + e]; }
            f16x8 ah, al, bh, bl;
            split_regs(a8, ah, al);
            split_regs(b8, bh, bl);
            ma[t] = mma3(ah, al, bh, bl, ma[t]);
        }
        // out = (Q KV) Z S, Z = 1 / (Q . ksum + eps)  (:46, :50); registers r < 8 belong to head 2 t, the others to 2 t + 1
        const float z0 = fS / (den0 + attn_eps), z1 = fS / (den1 + attn_eps);
#pragma unroll
        for (int r = 0; r < 16; ++r) ma[t][r] = ma[t][r] * (r < 8 ? z0 : z1) * (1.0f / ACT_SCALE);   // 2^-8 of the operands, x 2^4 for the split
    }

    // ------------------------------------------------------------------ merge, TRANSPOSED: D[m = channel][n = token]   (phases 12..15)
    f32x16 mg[CT];
    zero(mg);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        begin_phase(12 + t);
        request_w(14 + t);
        const unsigned char* slab = ring + ((12 + t) % RING) * SLAB + lane * 16;
        f16x8 ah[2], al[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float a8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a8[e] = ma[t][8 * u + e];
            split_regs(a8, ah[u], al[u]);
        }
        f16x8 bh[3], bl[3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bh[i] = *reinterpret_cast<const f16x8*>(slab + (i * 2) * 1024);
            bl[i] = *reinterpret_cast<const f16x8*>(slab + (i * 2 + 1) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2 * CT; ++i) {                     // i = u * CT + ct
            if (i + 2 < 2 * CT) {
                bh[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + ((i + 2) * 2) * 1024);
                bl[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + ((i + 2) * 2 + 1) * 1024);
            }
            const int u = i / CT, ct = i % CT;
            mg[ct] = mma3(bh[i % 3], bl[i % 3], ah[u], al[u], mg[ct]);        // weight fragment as A, message^T as B
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
    }

    // Activation-range guard (as K9 / K14): any operand beyond the split's range reaches `mg` as inf / NaN.
    float chk_all = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) chk_all += mg[ct][r];
    // ------------------------------------------------------------------ norm1: lane = token, registers = 64 of its 128 channels
    // mg[ct][r]: token l31, channel 32 ct + mfma32_row(r, h)
    const float inv_c = 1.0f / (float)DM;
    {
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) { mg[ct][r] *= sc.m; sum += mg[ct][r]; }
        sum += shfl_xor_f(sum, 32);
        const float mean = sum * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) { mg[ct][r] -= mean; sq += mg[ct][r] * mg[ct][r]; }
        sq += shfl_xor_f(sq, 32);
        const float rstd = 1.0f / sqrtf(sq * inv_c + ln_eps);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                   // registers 4 g4 .. 4 g4 + 3: four consecutive channels
                const int c0 = 32 * ct + 8 * g4 + 4 * h;
                const float4 gv = *reinterpret_cast<const float4*>(gamma + c0), bv = *reinterpret_cast<const float4*>(beta + c0);
                mg[ct][4 * g4 + 0] = mg[ct][4 * g4 + 0] * rstd * gv.x + bv.x;
                mg[ct][4 * g4 + 1] = mg[ct][4 * g4 + 1] * rstd * gv.y + bv.y;
                mg[ct][4 * g4 + 2] = mg[ct][4 * g4 + 2] * rstd * gv.z + bv.z;
                mg[ct][4 * g4 + 3] = mg[ct][4 * g4 + 3] * rstd * gv.w + bv.w;
                __builtin_amdgcn_sched_barrier(0);             // eight registers of gamma / beta at a time (hoisted, the 32 loads spill)
            }
    }

    // ================================================================== the MLP block on this wave's tile, in two halves of the
    // hidden dimension (4 of the 8 hidden tiles each: 64 accumulator registers next to the 64 of the message and the 64 of the
    // output -- with all 8 tiles at once the kernel spilled).  Per half: GEMM 1 transposed, H^T[hidden][token] = W0 [x | msg]^T,
    // message slabs first (B = the registers above), then the x slabs (B = x chunks, requested again by LDS-DMA); then GEMM 2 of
    // those four hidden tiles, D[m = token][n = channel] += relu(H) W2^T.  Slab = two k-steps x four hidden tiles (K14's shape).
    f32x16 acc2[CT];
    zero(acc2);
    float hid_chk = 0.f;
    auto gemm1_slab = [&](int s, f32x16 (&acc)[CT], const f16x8 (&bh2)[2], const f16x8 (&bl2)[2]) {
        const unsigned char* slab = ring + (s % RING) * SLAB + lane * 16;
        f16x8 wh[3], wl[3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wh[i] = *reinterpret_cast<const f16x8*>(slab + i * 2048);
            wl[i] = *reinterpret_cast<const f16x8*>(slab + i * 2048 + 1024);
        }
#pragma unroll
        for (int i = 0; i < 2 * CT; ++i) {                     // i = k-step * CT + tile
            if (i + 2 < 2 * CT) {
                wh[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + (i + 2) * 2048);
                wl[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + (i + 2) * 2048 + 1024);
            }
            const int ks = i / CT, t = i % CT;
            acc[t] = mma3(wh[i % 3], wl[i % 3], bh2[ks], bl2[ks], acc[t]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
    };
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int s0 = 16 + 12 * hh;                           // this half's first slab
        f32x16 acc1[CT];
        zero(acc1);
        if (true) request_x(4);                                // x chunk 0 (its 4 KiB is free: last read in the q phases / the previous half)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {                      // message slabs: both k-steps of message tile ct
            begin_phase(s0 + ct);
            if (s0 + ct + 2 < NSLAB) request_w(s0 + ct + 2);
            f16x8 bh2[2], bl2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float b8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) b8[e] = mg[ct][8 * u + e] * ACT_SCALE;
                split_regs(b8, bh2[u], bl2[u]);
            }
            gemm1_slab(s0 + ct, acc1, bh2, bl2);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {                          // x slabs: chunk c
            begin_phase(s0 + 4 + c);
            read_chunk(c + 5);                                 // x chunk c; next: x chunk c + 1 (none after the last)
            if (s0 + 4 + c + 2 < NSLAB) request_w(s0 + 4 + c + 2);
            gemm1_slab(s0 + 4 + c, acc1, xh, xl);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {                         // GEMM 2 of hidden tiles 4 hh + t
            const int s = s0 + 8 + t;
            begin_phase(s);
            if (s + 2 < NSLAB) request_w(s + 2);
            const unsigned char* slab = ring + (s % RING) * SLAB + lane * 16;
            f16x8 ha[2], hl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float a8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    hid_chk += acc1[t][8 * u + e];            // (ReLU's max would swallow a NaN)
                    a8[e] = fmaxf(acc1[t][8 * u + e] * hscale, 0.f);
                }
                split_regs(a8, ha[u], hl[u]);
            }
            f16x8 bh[3], bl[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bh[i] = *reinterpret_cast<const f16x8*>(slab + (i * 2) * 1024);
                bl[i] = *reinterpret_cast<const f16x8*>(slab + (i * 2 + 1) * 1024);
            }
#pragma unroll
            for (int i = 0; i < 2 * CT; ++i) {                 // i = u * CT + ct
                if (i + 2 < 2 * CT) {
                    bh[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + ((i + 2) * 2) * 1024);
                    bl[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + ((i + 2) * 2 + 1) * 1024);
                }
                const int u = i / CT, ct = i % CT;
                acc2[ct] = mma3(ha[u], hl[u], bh[i % 3], bl[i % 3], acc2[ct]);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
        }
    }
    if (overflow) {
        float chk = chk_all + hid_chk;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk += acc2[ct][r];
        if (__any(!(fabsf(chk) <= FLT_MAX)) && (threadIdx.x & 63) == 0) atomicOr(overflow, 1);
    }
    // ------------------------------------------------------------------ norm2 (DPP sums: lane = channel here) + x, store
    float g[CT], b[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { g[ct] = gamma2[32 * ct + l31]; b[ct] = beta2[32 * ct + l31]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v[CT];
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { v[ct] = acc2[ct][r] * oscale; sum += v[ct]; }
        const float mean = sum32(sum) * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { v[ct] -= mean; sq += v[ct] * v[ct]; }
        const float rstd = 1.0f / sqrtf(sum32(sq) * inv_c + ln2_eps);
        const int tok = mfma32_row(r, h);
        if (live && tok < L) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const long o = (win * L + tok) * DM + 32 * ct + l31;
                out[o] = x[o] + (v[ct] * rstd * g[ct] + b[ct]);
            }
        }
    }
}

}  // namespace

extern "C" {

size_t far_layer128_packed_bytes(int d_model) { return d_model == DM ? (size_t)NSLAB * SLAB : 0; }

// out [nwin][L][128] = x + norm2(mlp(cat[x, norm1(merge(LinearAttention(q_proj(x), k_proj(src), v_proj(src))))]))  -- a whole
// LoFTREncoderLayer (transformer.py:44-67) at d_model = 128, 8 heads of 16, sequences of at most 32 tokens (the fine-level windows),
// no masks.  x [nwin][L][128], src [nwin][S][128] fp32.  packed: far_amd/ops.py:PackedLayer128 (K14's image, then the MLP's with
// W0's message columns in accumulator order); scale_* = 2^-(w_exp + 4) of Wk, Wv, Wq, Wm; hscale / oscale as far_mlp_fused_f16s;
// gamma / beta / ln_eps = norm1, gamma2 / beta2 / ln2_eps = norm2.  out must not alias x / src.
int far_layer128_f16s(const float* x, const float* src, const void* packed, long nwin, int L, int S, int d_model, int heads,
                      float scale_k, float scale_v, float scale_q, float scale_m, float attn_eps, const float* gamma,
                      const float* beta, float ln_eps, float hscale, float oscale, const float* gamma2, const float* beta2,
                      float ln2_eps, float* out, int* overflow, hipStream_t stream) {
    far_clear_errors();
    if (nwin == 0) return FAR_OK;
    if (!x || !src || !packed || !gamma || !beta || !gamma2 || !beta2 || !out || nwin < 0 || L <= 0 || S <= 0 || L > 32 || S > 32 ||
        d_model != DM || heads != 8 || out == x || out == src)
        return FAR_EINVAL;
    const long nb = (nwin + WAVES - 1) / WAVES;
    if (nb > 0x7fffffffL) return FAR_EINVAL;
    const Scales sc{scale_k, scale_v, scale_q, scale_m};
    hipLaunchKernelGGL(k_layer128, dim3((unsigned)nb), dim3(64 * WAVES), 0, stream, x, src, (const unsigned char*)packed, nwin, L, S,
                       sc, attn_eps, gamma, beta, ln_eps, hscale, oscale, gamma2, beta2, ln2_eps, out, overflow);
    return far_check_launch();
}

}  // extern "C"
