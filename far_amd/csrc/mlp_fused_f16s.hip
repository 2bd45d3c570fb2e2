// K13: the MLP block of a LoFTR encoder layer at d_model = 128 (the fine-level transformer) as ONE kernel.
//
// Replaces mp3d_loftr/src/loftr/loftr_module/transformer.py:64-67 (LoFTREncoderLayer.forward, second half):
//     message = self.mlp(torch.cat([x, message], dim=2))      mlp = Linear(2d, 2d, no bias) -> ReLU -> Linear(2d, d, no bias)
//     message = self.norm2(message)
//     return x + message
// The fine level runs this on 61 k windows x 25 tokens x 128 channels per 32 pairs, four times per step; as two K9
// launches it moves the 256-channel hidden tensor to HBM and back (3.1 GB of the 6.3 GB the two launches touch) and the
// stage is bandwidth-bound.  Here the hidden activations never leave the registers:
//   GEMM 1, transposed:  H^T[hc][row] = sum_k W0[hc][k] X[row][k]      A = W0 fragment (LDS), B = the wave's own 32 rows
//                        (read from global as 64 contiguous bytes per lane and chunk, split in registers)
//   ReLU, x 2^4, split into fp16 (hi, lo) in the accumulator registers: a lane owns one row and, per accumulator tile,
//   the hidden channels 16 u + 4 h + (e & 3) + 8 (e >> 2) -- which is exactly an A operand A[m = row][k] of
//   GEMM 2:              Y[row][c] = sum_hc H[row][hc] W2[c][hc]        B = W2 fragment (LDS), packed in that k order
//   LayerNorm over the 128 channels of a row inside the wave (DPP sums over the 32 lanes of a half-wave), + x, store.
// Arithmetic as K9: every product as three f16 MFMAs on (hi, lo) operand pairs, fp32 accumulation (fp32-grade).
// Workgroup = 4 waves = 128 rows; its waves consume the same 24 weight slabs of 16 KiB (16 k-steps of W0, 8 hidden tiles
// of W2) from a 3-slot LDS ring filled by asm LDS-DMA two slabs ahead (one barrier per slab = per 24 MFMAs of a wave); the
// input rows arrive by LDS-DMA too (4 KiB per wave), so every memory request of the loop is waited for by hand.
// 64 KiB LDS, <= 256 VGPRs: two independent workgroups per CU (one workgroup of eight waves moved in lock step with its
// barrier and left the matrix pipe idle during every LDS round trip: 1.18 ms against 1.47 ms for the two K9 launches).
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int DM = 128;                  // d_model
constexpr int KIN = 2 * DM;              // input channels of GEMM 1: [x | message]
constexpr int HID = 2 * DM;              // hidden channels
constexpr int NS1 = KIN / 16;            // k-steps of GEMM 1 (one slab each)
constexpr int HT = HID / 32;             // hidden tiles (one W2 slab each: two k-steps)
constexpr int CT = DM / 32;              // output tiles
constexpr int SLAB = 16384;              // bytes per slab: GEMM 1 [HT][plane][1 KiB]; GEMM 2 [u][CT][plane][1 KiB]
constexpr int NSLAB = NS1 + HT;          // 24
constexpr int RING = 3;
constexpr int WAVES = 4;
constexpr int PIECES = 16 / WAVES;       // 1 KiB DMA requests per wave and slab
constexpr float ACT_SCALE = 16.0f;       // activations x 2^4 before the split (as K9)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

__device__ __forceinline__ void split8(const float4& u, const float4& v, float scale, f16x8& hi, f16x8& lo) {
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 8; i += 2) {                       // packed conversions (common.h: split2)
        f16x2 h, l;
        split2(f32x2{x[i], x[i + 1]} * f32x2{scale, scale}, h, l);
        hi[i] = h.x; hi[i + 1] = h.y;
        lo[i] = l.x; lo[i + 1] = l.y;
    }
}

#define FAR_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
// sum over the 32 lanes of each half-wave, result in every lane
__device__ __forceinline__ float sum32(float v) {
    v += FAR_DPP_F(v, 0xB1);       // quad_perm [1,0,3,2]
    v += FAR_DPP_F(v, 0x4E);       // quad_perm [2,3,0,1]
    v += FAR_DPP_F(v, 0x141);      // row_half_mirror
    v += FAR_DPP_F(v, 0x140);      // row_mirror
    return v + shfl_xor_f(v, 16);
}

// Every phase waits for all of this wave's outstanding requests (vmcnt(0)) AND for its LDS reads (lgkmcnt(0): ring_barrier,
// common.h) before the barrier.  The requests are issued two slabs / one chunk ahead, so the youngest is a phase old (measured:
// no slower than counted waits).  Until round 6 the lgkmcnt(0) was missing: hipcc sinks the last MFMAs of a slab below the asm
// barrier, a wave crossed it with two fragment reads queued and a sibling's re-request of the slot could land first -- 1 437
// windows of 120 296 wrong over 20 launches next to a busy second stream (profiles/r06_ring_race.txt), none with the wait.

// SPLIT = false (far_mlp_fused_f16, round 5): plain fp16 operands, one MFMA per product; the data movement is unchanged
template <bool SPLIT, bool LGKM = true>
__global__ __launch_bounds__(256, 2) void k_mlp128(const float* __restrict__ x, const float* __restrict__ msg,
                                                   const unsigned char* __restrict__ wimg, long R, float hscale, float oscale,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                   float* __restrict__ out, int* __restrict__ overflow) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[RING * SLAB];
    __shared__ __attribute__((aligned(16))) unsigned char xs[WAVES * 4096];          // per wave: its 32 rows x 128 B of the current chunk
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: it enters the M0 values of the LDS-DMA requests
    const long row0 = (long)blockIdx.x * (32 * WAVES) + 32 * wave;
    const unsigned ring_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)ring);
    const unsigned xs_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)(xs + wave * 4096));
    const unsigned char* wsrc = wimg + (size_t)lane * 16;
    auto request_w = [&](int s) {                             // slab s -> ring slot s % RING: pieces wave, wave + 4, ...
        const unsigned dst = ring_base + (unsigned)((s % RING) * SLAB);
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            glds16(wsrc + (size_t)s * SLAB + (wave + WAVES * i) * 1024, dst + (wave + WAVES * i) * 1024);
    };
    // Input chunk c (32 channels: 0..3 from x, 4..7 from the message) of this wave's 32 rows -> its private 4 KiB of LDS,
    // row-major [32 rows][8 pieces of 16 B], piece q of row r holding source piece q ^ ((r ^ (r >> 3)) & 7) (source-side
    // swizzle: the ds_read_b128 below are conflict free).  Request j moves rows 8 j .. 8 j + 7: lane = (row & 7, piece).
    const int rr = lane >> 3, q = lane & 7;
    auto request_x = [&](int c) {
        const float* base = (c < 4 ? x : msg) + 32 * (c & 3);
        const unsigned dst = xs_base;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 8 * j + rr;
            long gr = row0 + r;
            gr = gr < R ? gr : R - 1;                          // rows past the end: any valid row (never stored)
            glds16(base + gr * DM + 4 * (q ^ ((r ^ (r >> 3)) & 7)), dst + j * 1024);
        }
    };
    request_x(0);
    request_w(0);
    request_w(1);
    f16x8 xh[2], xl[2];
    f32x16 acc1[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
    // this lane's row l31, channels 16 h .. 16 h + 15 of a chunk = source pieces 4 h .. 4 h + 3
    const unsigned char* xrd = xs + wave * 4096 + (l31 >> 3) * 1024 + (l31 & 7) * 128;
    const int sw = (l31 ^ (l31 >> 3)) & 7;

    // ---------------------------------------------------------------- GEMM 1: 16 k-steps, one slab each
#pragma unroll
    for (int s = 0; s < NS1; ++s) {                            // unrolled: ring slots and chunk parity are immediates
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier<LGKM>();                                                 // all 16 pieces of slab s visible; slot (s + 2) % 3 is free
        if ((s & 1) == 0) {                                   // a new chunk: read this lane's 64 bytes, split
            float4 raw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) raw[i] = *reinterpret_cast<const float4*>(xrd + (((4 * h + i) ^ sw) * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read out before the next chunk is requested into the same 4 KiB
            if ((s >> 1) + 1 < KIN / 32) request_x((s >> 1) + 1);
            split8(raw[0], raw[1], ACT_SCALE, xh[0], xl[0]);
            split8(raw[2], raw[3], ACT_SCALE, xh[1], xl[1]);
        }
        request_w(s + 2);                                     // s + 2 <= 17 < NSLAB
        const unsigned char* slab = ring + (s % RING) * SLAB + lane * 16;
        const f16x8 bh = xh[s & 1], bl = xl[s & 1];
        // fragment reads run two tiles ahead of their MFMAs (the two waves of a SIMD belong to this one workgroup and move
        // in step with the per-slab barrier: nobody else hides an exposed LDS round trip)
        f16x8 ah[3], al[3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = *reinterpret_cast<const f16x8*>(slab + t * 2048);
            al[t] = *reinterpret_cast<const f16x8*>(slab + t * 2048 + 1024);
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            if (t + 2 < HT) {
                ah[(t + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + (t + 2) * 2048);
                al[(t + 2) % 3] = *reinterpret_cast<const f16x8*>(slab + (t + 2) * 2048 + 1024);
            }
            acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t % 3], bh, acc1[t], 0, 0, 0);
            if (SPLIT) {
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t % 3], bl, acc1[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t % 3], bh, acc1[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, SPLIT ? 2 : 1, 0);     // the reads of tile t + 2
            __builtin_amdgcn_sched_group_barrier(0x008, SPLIT ? 3 : 1, 0);     // the MFMAs of tile t
        }
    }

    // ---------------------------------------------------------------- GEMM 2: 8 hidden tiles, one slab (two k-steps) each
    f32x16 acc2[CT];
    float hid_chk = 0.f;                  // activation-range guard: sum of the hidden tensor's raw accumulators
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[ct][r] = 0.f;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const int s = NS1 + t;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier<LGKM>();
        if (s + 2 < NSLAB) request_w(s + 2);
        const unsigned char* slab = ring + (s % RING) * SLAB + lane * 16;
        // hidden = relu(acc * 2^-(w_exp + 4)), then x 2^4 for the split: hscale = 2^-w_exp.  Both k-steps' A operands first
        // (VALU), then the eight (k-step, output tile) products with their W2 fragments read two ahead.
        f16x8 ha[2], hl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                hid_chk += acc1[t][8 * u + e] + acc1[t][8 * u + e + 1];       // (ReLU's max would swallow a NaN: inf - inf of an overflowed input)
                const f32x2 hv = __builtin_elementwise_max(f32x2{acc1[t][8 * u + e], acc1[t][8 * u + e + 1]} * f32x2{hscale, hscale}, f32x2{0.f, 0.f});
                f16x2 h2, l2;
                split2(hv, h2, l2);
                ha[u][e] = h2.x; ha[u][e + 1] = h2.y;
                hl[u][e] = l2.x; hl[u][e + 1] = l2.y;
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
            acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[u], bh[i % 3], acc2[ct], 0, 0, 0);
            if (SPLIT) {
                acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[u], bl[i % 3], acc2[ct], 0, 0, 0);
                acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hl[u], bh[i % 3], acc2[ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, SPLIT ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, SPLIT ? 3 : 1, 0);
        }
    }

    // Activation-range guard (as K9): an input or a hidden value beyond the fp16 range of its 2^4-scaled split (|a| > 4094)
    // is inf in the operand and inf / NaN here; the sum of the wave's accumulators is finite exactly when all of them are.
    if (overflow) {
        float chk = hid_chk;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk += acc2[ct][r];
        if (__any(!(fabsf(chk) <= FLT_MAX)) && (threadIdx.x & 63) == 0) atomicOr(overflow, 1);
    }
    // ---------------------------------------------------------------- LayerNorm (two-pass, as K6 / K9) + residual + store
    // acc2[ct][r]: row row0 + mfma32_row(r, h), channel 32 ct + l31
    const float inv_c = 1.0f / (float)DM;
    float g[CT], b[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { g[ct] = gamma[32 * ct + l31]; b[ct] = beta[32 * ct + l31]; }
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
        const float rstd = 1.0f / sqrtf(sum32(sq) * inv_c + eps);
        const long orow = row0 + mfma32_row(r, h);
        if (orow < R) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const long o = orow * DM + 32 * ct + l31;
                out[o] = x[o] + (v[ct] * rstd * g[ct] + b[ct]);
            }
        }
    }
}

template <bool SPLIT>
int mlp_fused_launch(const float* x, const float* msg, const void* packed, long R, int d_model, float hscale, float oscale,
                     const float* gamma, const float* beta, float eps, float* out, int* overflow, hipStream_t stream) {
    far_clear_errors();
    if (R == 0) return FAR_OK;
    if (!x || !msg || !packed || !gamma || !beta || !out || R < 0 || d_model != DM || out == x || out == msg) return FAR_EINVAL;
    const long nb = (R + 32 * WAVES - 1) / (32 * WAVES);
    if (nb > 0x7fffffffL) return FAR_EINVAL;
#ifdef FAR_RING_EXP
    if (far_get_tuning(11) & 4) {        // experiment build: the barrier without its lgkmcnt(0) (rounds 3-5)
        hipLaunchKernelGGL((k_mlp128<SPLIT, false>), dim3((unsigned)nb), dim3(64 * WAVES), 0, stream, x, msg, (const unsigned char*)packed, R,
                           hscale, oscale, gamma, beta, eps, out, overflow);
        return far_check_launch();
    }
#endif
    hipLaunchKernelGGL(k_mlp128<SPLIT>, dim3((unsigned)nb), dim3(64 * WAVES), 0, stream, x, msg, (const unsigned char*)packed, R, hscale,
                       oscale, gamma, beta, eps, out, overflow);
    return far_check_launch();
}

}  // namespace

extern "C" {

// bytes of the packed weight image of far_mlp_fused_f16s (24 slabs of 16 KiB; layout in far_amd/ops/fine.py:PackedMlp)
size_t far_mlp_fused_packed_bytes(int d_model) { return d_model == DM ? (size_t)NSLAB * SLAB : 0; }

// out [R][128] = x + LayerNorm(W2 relu(W0 [x | msg]))   (transformer.py:64-67 at d_model = 128)
//   x, msg [R][128] fp32; packed: the image PackedMlp builds (W0 scaled by 2^w0_exp, W2 by 2^w2_exp, fp16 hi / lo planes in
//   execution order); hscale = 2^-w0_exp (accumulator -> 2^4 x hidden), oscale = 2^-(w2_exp + 4) (accumulator -> output);
//   gamma, beta [128], eps: norm2.  out may not alias x or msg.
int far_mlp_fused_f16s(const float* x, const float* msg, const void* packed, long R, int d_model, float hscale, float oscale,
                       const float* gamma, const float* beta, float eps, float* out, int* overflow, hipStream_t stream) {
    return mlp_fused_launch<true>(x, msg, packed, R, d_model, hscale, oscale, gamma, beta, eps, out, overflow, stream);
}

// The same block on plain fp16 operands (one MFMA per product; the 16-bit-operand class, LoFTR.set_precision('fp16')): same
// arguments, same packed image (its lo planes are loaded and not used), same activation-range flag.
int far_mlp_fused_f16(const float* x, const float* msg, const void* packed, long R, int d_model, float hscale, float oscale,
                      const float* gamma, const float* beta, float eps, float* out, int* overflow, hipStream_t stream) {
    return mlp_fused_launch<false>(x, msg, packed, R, d_model, hscale, oscale, gamma, beta, eps, out, overflow, stream);
}

}  // extern "C"
