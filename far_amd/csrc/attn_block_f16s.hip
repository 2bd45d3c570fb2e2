// K14: the attention block of a LoFTR encoder layer at d_model = 128 on short sequences (the 25-token fine windows) as
// ONE kernel: q / k / v projections, linear attention, merge, norm1.
//
// Replaces mp3d_loftr/src/loftr/loftr_module/transformer.py:51-61 (LoFTREncoderLayer.forward, first half)
//     query, key, value = q_proj(x), k_proj(source), v_proj(source)            [N, L|S, 8 heads, 16]
//     message = self.attention(query, key, value)                              linear_attention.py:31-50
//     message = self.norm1(self.merge(message.view(bs, -1, 128)))
// with  Q = elu(q) + 1, K = elu(k) + 1, KV = sum_s K_s (x) (v_s / S), Z = 1 / (Q . sum_s K_s + eps),  out = (Q KV) Z S.
// As separate launches (three K9 Linear layers, K5, K9 merge + LayerNorm) this is 11 passes over 781 MB tensors per
// layer-side at the fine level (61 k windows x 25 tokens per 32 pairs) and bandwidth-bound; here x and source are read
// once and the normalised message written once.
//
// One wave = one window, padded to a 32-row MFMA tile (rows >= L / S are masked), so the attention never leaves the
// wave; workgroup = 4 waves sharing the weight slabs (16 KiB each, 3-slot LDS ring, asm LDS-DMA TWO phases ahead; the wave's
// input rows arrive by LDS-DMA as well, double-buffered), two workgroups per CU (round 6; round 5 shipped 8 waves / 4 slots / one
// workgroup per CU -- still selectable, far_set_tuning(11, 1)).
// Every request of the loop is waited for by hand with a COUNTED vmcnt (table VM_ALLOW below): a phase waits only for the
// slab and the chunk it reads, 4-8 younger requests stay in flight across its barrier.  The round-3 form drained the queue
// (vmcnt(0)) in front of each of its 16 barriers with requests only one to two phases old: 2.7 us per phase against 0.4 us
// of matrix work -- the kernel ran at the memory latency (profiles/r05_step_floors_baseline.txt: 3.4x its floors).  The GEMMs alternate
// between the two accumulator orientations so that each result is already the operand of its consumer:
//   k, v   D[m = token][n = channel]  (lane = channel): K'^T V per 32-channel tile (two heads; the cross-head quarters of
//          the 32 x 32 block are zeroed) contracts over tokens = over the REGISTERS of a lane -> both are MFMA operands
//          as they stand;  sum_s K'_s is an in-lane sum
//   q^T    D[m = channel][n = token]  (lane = token): B operand of  message^T = KV^T Q'^T  (A = the KV block, lane = e)
//   message^T (lane = token, registers = channels)  = the A operand of the merge GEMM (weights packed in that k order)
//   merge  D[m = token][n = channel]: LayerNorm over the channels = DPP sums over the 32 lanes of a half-wave, store.
// Arithmetic as K9 / K13: every product three f16 MFMAs on (hi, lo) pairs, fp32 accumulation.
#include "common.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int DM = 128;                  // d_model
constexpr int CT = DM / 32;              // 32-channel tiles (two 16-channel heads each)
constexpr int SLAB = 16384;              // two k-steps x 4 tiles x 2 planes x 1 KiB
constexpr int NSLAB = 16;                // [k c0][v c0][k c1][v c1][k c2][v c2][k c3][v c3][q c0..c3][merge t0..t3]
constexpr int RING = 4;                  // the 8-wave form (far_set_tuning(11, 1)); the default form: 3 slots, 4 waves (attn_block_launch)
constexpr int WAVES = 8;
constexpr int XBUF = 4096;               // one 32-channel chunk of a window's rows
constexpr int SMEM_NEW = RING * SLAB + WAVES * 2 * XBUF;      // 64 KiB + 64 KiB
// Requests of one wave in program order (counted pipeline): prologue X0 X1 W0 .. W(D-1); phase p (after its barrier): [X(j + 2) if the
// phase read chunk j and j + 2 < 8] then [W(p + D) if p + D < 16]; X = 4 pieces, W = 16 / waves pieces; D = ring slots - 1.  Chunks are read
// in phases 0, 2, 4, 6 (source) and 8..11 (x).  vm_allow(p) = pieces issued after the youngest request phase p reads (its slab W(p),
// its chunk) = what may stay in flight across barrier p -- computed by replaying exactly this schedule (4 waves, 3 slots:
// 4 8 4 8 4 8 4 8 4 8 8 4 4 4 4 0; 8 waves, 4 slots: 4 8 8 8 8 8 8 8 8 8 8 4 4 4 2 0).  A wrong entry shows as stale LDS reads: the
// float64 tests of tests/test_attn_block_gpu.py and the bench-scale determinism tests of tests/test_determinism_gpu.py catch that.
constexpr int chunk_read_in_phase(int p) { return (p < 8 && (p & 1) == 0) ? p / 2 : ((p >= 8 && p < 12) ? 4 + (p - 8) : -1); }
constexpr int vm_allow(int p, int wp, int d) {
    // kind 0 = X(id), 1 = W(id)
    int kind[64] = {}, id[64] = {}, n[64] = {}, cnt = 0;
    kind[cnt] = 0; id[cnt] = 0; n[cnt++] = 4;
    kind[cnt] = 0; id[cnt] = 1; n[cnt++] = 4;
    for (int s0 = 0; s0 < d; ++s0) { kind[cnt] = 1; id[cnt] = s0; n[cnt++] = wp; }
    for (int q = 0; q < p; ++q) {
        const int j = chunk_read_in_phase(q);
        if (j >= 0 && j + 2 < 8) { kind[cnt] = 0; id[cnt] = j + 2; n[cnt++] = 4; }
        if (q + d < NSLAB) { kind[cnt] = 1; id[cnt] = q + d; n[cnt++] = wp; }
    }
    const int jx = chunk_read_in_phase(p);
    int last = -1;
    for (int i = 0; i < cnt; ++i)
        if ((kind[i] == 1 && id[i] == p) || (kind[i] == 0 && id[i] == jx)) last = i;
    int after = 0;
    for (int i = last + 1; i < cnt; ++i) after += n[i];
    return after;
}
struct VmTable {
    int a[NSLAB];
    constexpr VmTable(int wp, int d) : a{} {
        for (int p = 0; p < NSLAB; ++p) a[p] = vm_allow(p, wp, d);
    }
};
static_assert(vm_allow(0, 2, 3) == 4 && vm_allow(1, 2, 3) == 8 && vm_allow(10, 2, 3) == 8 && vm_allow(11, 2, 3) == 4 &&
              vm_allow(14, 2, 3) == 2 && vm_allow(15, 2, 3) == 0, "the 8-wave / 4-slot schedule");
static_assert(vm_allow(0, 4, 2) == 4 && vm_allow(1, 4, 2) == 8 && vm_allow(2, 4, 2) == 4 && vm_allow(9, 4, 2) == 8 && vm_allow(10, 4, 2) == 8 &&
              vm_allow(11, 4, 2) == 4 && vm_allow(14, 4, 2) == 4 && vm_allow(15, 4, 2) == 0, "the 4-wave / 3-slot schedule");
constexpr float ACT_SCALE = 16.0f;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
// F.elu(x) + 1 = x + 1 (x > 0), e^x otherwise, as 2^(x log2 e) on the hardware exponential (as K9's LinearAttention epilogues,
// conv_igemm_f16s.hip:la_elu1): the rounding of the argument costs |x| e^x 2^-24 <= 2^-25 absolute -- half an ulp of 1.0, the size
// of a K' -- where expm1f(x) + 1 rounds twice; 4 instructions instead of ~37 and two branches per value (round 5: the 128 expm1f
// expansions were half of this kernel's 10.8 k instructions per window).
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

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
// SPLIT = false (far_attn_block_f16, round 5): plain fp16 operands -- the hi.hi product only; the data movement is unchanged
template <bool SPLIT>
__device__ __forceinline__ f32x16 mma3(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    if (!SPLIT) return c;
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

// vmcnt(n) with an immediate: the phase loops are fully unrolled, so `n` is a constant and the switch folds away
__device__ __forceinline__ void wait_vm(int n) {
#define FAR_VM_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        FAR_VM_CASE(0) FAR_VM_CASE(1) FAR_VM_CASE(2) FAR_VM_CASE(3) FAR_VM_CASE(4) FAR_VM_CASE(5) FAR_VM_CASE(6) FAR_VM_CASE(7) FAR_VM_CASE(8)
        FAR_VM_CASE(9) FAR_VM_CASE(10) FAR_VM_CASE(11) FAR_VM_CASE(12) FAR_VM_CASE(13) FAR_VM_CASE(14) FAR_VM_CASE(15) FAR_VM_CASE(16)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef FAR_VM_CASE
}

// NW waves per workgroup, NR ring slots; CNT: counted waits, requests NR - 1 phases ahead, double-buffered input chunks (!CNT: every
// phase drains the queue, one buffer).  History of the pipeline, because it cost two rounds: the round-3 form <4, 3, false> and a
// counted <4, 3, true> -- both TWO workgroups per CU -- were 4-5 % faster than <8, 4, true> but differed run to run in round 5 once
// the elu got cheap (5 and 156 windows of 120 296 per launch); round 5 shipped the 8-wave form "because it passed".  Round 6 found
// the cause (common.h: ring_barrier; docs/rounds/r06.md section 1): the barrier lacked `s_waitcnt lgkmcnt(0)`, hipcc sank each phase's
// last MFMAs and their lgkmcnt wait below it, and a sibling's re-request of the slot could overtake the two fragment reads still
// queued.  EVERY form had the hole -- next to a busy second stream the shipped 8-wave form differed in 58 windows over 20 launches,
// K13 in 1 437 (tools/ring_ab.py, profiles/r06_ring_race.txt); with the wait in place all forms are bit-identical over 20 launches
// of 120 296 windows next to a busy stream, and the faster two-workgroups-per-CU counted form ships again.
template <int NW, int NR, bool CNT, bool SPLIT, bool LGKM = true>
__global__ __launch_bounds__(64 * NW, 2) void k_attn128(const float* __restrict__ x, const float* __restrict__ src,
                                                        const unsigned char* __restrict__ wimg, long nwin, int L, int S, Scales sc,
                                                        float attn_eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float ln_eps, float* __restrict__ out, int* __restrict__ overflow) {
    constexpr int NPIECE = 16 / NW;                            // 1 KiB pieces of a slab per wave
    constexpr int AHEAD = NR - 1;                              // slab p + AHEAD is requested in phase p
    constexpr int XB = CNT ? 2 : 1;                            // input-chunk buffers per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const xs = smem + NR * SLAB;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long win = (long)blockIdx.x * NW + wave;
    const bool live = win < nwin;
    if (!live) win = nwin - 1;                                 // a spare wave of the last workgroup: works on a valid window, stores nothing
    const unsigned ring_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)ring);
    const unsigned xs_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)(xs + wave * (XB * XBUF)));
    const unsigned char* wsrc = wimg + (size_t)lane * 16;
    auto request_w = [&](int s) {
        const unsigned dst = ring_base + (unsigned)((s % NR) * SLAB);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i)
            glds16(wsrc + (size_t)s * SLAB + (wave + NW * i) * 1024, dst + (wave + NW * i) * 1024);
    };
    // chunk j: 32 channels of this window's rows -> 4 KiB of the wave (buffer j % XB), row-major [32 rows][8 pieces of 16 B],
    // source-side swizzle (piece q of row r holds source piece q ^ ((r ^ (r >> 3)) & 7)).  j < 4: source channels 32 j; j >= 4: x.
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
            glds16(base + gr * DM + 4 * (q8 ^ ((r ^ (r >> 3)) & 7)), xs_base + (j % XB) * XBUF + i * 1024);
        }
    };
    request_x(0);
    if (CNT) request_x(1);
#pragma unroll
    for (int s0 = 0; s0 < AHEAD; ++s0) request_w(s0);
    const unsigned char* xrd = xs + wave * (XB * XBUF) + (l31 >> 3) * 1024 + (l31 & 7) * 128;
    const int sw = (l31 ^ (l31 >> 3)) & 7;
    f16x8 xh[2], xl[2];                                       // the current chunk of this lane's row: two k-steps of (hi, lo)
    auto read_chunk = [&](int j) {                             // read out the landed chunk j, then request the chunk that reuses its buffer
        float4 raw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) raw[i] = *reinterpret_cast<const float4*>(xrd + (j % XB) * XBUF + (((4 * h + i) ^ sw) * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j + XB < 8) request_x(j + XB);
        split8(raw[0], raw[1], ACT_SCALE, xh[0], xl[0]);
        split8(raw[2], raw[3], ACT_SCALE, xh[1], xl[1]);
    };
    // one slab of a projection GEMM: two k-steps x four 32-channel tiles.  TRANSPOSED: D[m = channel][n = token] (weights are
    // the A operand), else D[m = token][n = channel]
    auto gemm_slab = [&](int s, f32x16 (&acc)[CT], auto transposed) {
        constexpr bool TR = decltype(transposed)::value;
        const unsigned char* slab = ring + (s % NR) * SLAB + lane * 16;
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
            if (TR) acc[t] = mma3<SPLIT>(wh[i % 3], wl[i % 3], xh[ks], xl[ks], acc[t]);
            else acc[t] = mma3<SPLIT>(xh[ks], xl[ks], wh[i % 3], wl[i % 3], acc[t]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
    };
    // phase p: slab p (and the chunk it reads) has landed for this wave -- then the barrier makes every wave's pieces visible and
    // frees the slot slab p - 1 used; after it the phase requests slab p + AHEAD into that slot
    auto begin_phase = [&](int p) {
        constexpr VmTable vmt(NPIECE, AHEAD);
        wait_vm(CNT ? vmt.a[p] : 0);
        ring_barrier<LGKM>();                                  // common.h: no LDS read of this wave in flight at the barrier
        __builtin_amdgcn_sched_barrier(0);
    };
    auto next_w = [&](int p) { if (p + AHEAD < NSLAB) request_w(p + AHEAD); };
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
        read_chunk(c);                                         // source chunk c; its buffer is re-requested (chunk c + 1 or c + 2; >= 4: x)
        next_w(2 * c);
        gemm_slab(2 * c, ka, std::false_type{});
        begin_phase(2 * c + 1);
        next_w(2 * c + 1);
        gemm_slab(2 * c + 1, va, std::false_type{});
    }
    // K' = elu(k) + 1, V = v / S on the rows of the window (registers: token mfma32_row(r, h); lane: channel 32 t + l31)
    float ksum[CT];
    const float fS = (float)S;
    const float v_mul = sc.v * ACT_SCALE / fS;                  // values / v_length (linear_attention.py:43) folded with the accumulator scale: one
                                                              // rounding of the factor (<= 1 ulp against the reference's two operations), no division per value
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        float s_ = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = mfma32_row(r, h) < S;
            const float kk = ok ? elu1(ka[t][r] * sc.k) : 0.f;
            ka[t][r] = kk * ACT_SCALE;
            va[t][r] = ok ? va[t][r] * v_mul : 0.f;
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
            kv[t] = mma3<SPLIT>(ah, al, bh, bl, kv[t]);
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
        read_chunk(4 + c);                                     // x chunk c
        next_w(8 + c);
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
            ma[t] = mma3<SPLIT>(ah, al, bh, bl, ma[t]);
        }
        // out = (Q KV) Z S, Z = 1 / (Q . ksum + eps)  (:46, :50); registers r < 8 belong to head 2 t, the others to 2 t + 1
        const float z0 = fS / (den0 + attn_eps), z1 = fS / (den1 + attn_eps);
#pragma unroll
        for (int r = 0; r < 16; ++r) ma[t][r] = ma[t][r] * (r < 8 ? z0 : z1) * (1.0f / ACT_SCALE);   // 2^-8 of the operands, x 2^4 for the split
    }

    // ------------------------------------------------------------------ merge: D[m = token][n = channel]   (phases 12..15)
    f32x16 mg[CT];
    zero(mg);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        begin_phase(12 + t);
        next_w(12 + t);
        const unsigned char* slab = ring + ((12 + t) % NR) * SLAB + lane * 16;
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
            mg[ct] = mma3<SPLIT>(ah[u], al[u], bh[i % 3], bl[i % 3], mg[ct]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
    }

    // Activation-range guard (as K9).  The operands of this kernel's chained products are 2^4-scaled activations AND their
    // products (K'^T V, then KV Q'): the fp16 split overflows when |x| > 4094, or when a head's sum_s K'_s V_s / S or a
    // message entry exceeds 4094 -- tighter than K9's per-activation bound.  Any such inf reaches `mg` as inf / NaN.
    if (overflow) {
        float chk = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk += mg[ct][r];
        if (__any(!(fabsf(chk) <= FLT_MAX)) && (threadIdx.x & 63) == 0) atomicOr(overflow, 1);
    }
    // ------------------------------------------------------------------ norm1 over the 128 channels of a token, store
    const float inv_c = 1.0f / (float)DM;
    float g[CT], b[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { g[ct] = gamma[32 * ct + l31]; b[ct] = beta[32 * ct + l31]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v[CT];
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { v[ct] = mg[ct][r] * sc.m; sum += v[ct]; }
        const float mean = sum32(sum) * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { v[ct] -= mean; sq += v[ct] * v[ct]; }
        const float rstd = 1.0f / sqrtf(sum32(sq) * inv_c + ln_eps);
        const int tok = mfma32_row(r, h);
        if (live && tok < L) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) out[(win * L + tok) * DM + 32 * ct + l31] = v[ct] * rstd * g[ct] + b[ct];
        }
    }
}

}  // namespace

extern "C" {

size_t far_attn_block_packed_bytes(int d_model) { return d_model == DM ? (size_t)NSLAB * SLAB : 0; }

}  // extern "C"

// out [nwin][L][128] = norm1(merge(LinearAttention(q_proj(x), k_proj(src), v_proj(src))))   (transformer.py:51-61 at d_model = 128,
// 8 heads of 16, sequences of at most 32 tokens: the fine-level windows).  x [nwin][L][128], src [nwin][S][128] fp32;
// packed: the image far_amd/ops/fine.py:PackedAttn builds; scale_* = 2^-(w_exp + 4) of Wk, Wv, Wq, Wm; out must not alias x / src.
template <bool SPLIT>
static int attn_block_launch(const float* x, const float* src, const void* packed, long nwin, int L, int S, int d_model, int heads,
                             float scale_k, float scale_v, float scale_q, float scale_m, float attn_eps, const float* gamma,
                             const float* beta, float ln_eps, float* out, int* overflow, hipStream_t stream) {
    far_clear_errors();
    if (nwin == 0) return FAR_OK;
    if (!x || !src || !packed || !gamma || !beta || !out || nwin < 0 || L <= 0 || S <= 0 || L > 32 || S > 32 ||
        d_model != DM || heads != 8 || out == x || out == src)
        return FAR_EINVAL;
    const Scales sc{scale_k, scale_v, scale_q, scale_m};
    auto launch = [&](auto kern, int waves, int smem) -> int {
        bool cfg_failed = false;             // (a generic lambda: one flag per kernel instantiation)
        FAR_ONCE_PER_DEVICE(cfg_failed = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess);
        if (cfg_failed) return far_check_launch();
        const long nb = (nwin + waves - 1) / waves;
        if (nb > 0x7fffffffL) return FAR_EINVAL;
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(64 * waves), smem, stream, x, src, (const unsigned char*)packed, nwin, L, S, sc,
                           attn_eps, gamma, beta, ln_eps, out, overflow);
        return far_check_launch();
    };
    // far_set_tuning(11, v): 0 (default) = <4 waves, 3 slots, counted waits>, two workgroups per CU; 1 = <8 waves, 4 slots, counted>,
    // one workgroup per CU (round 5's form); 2 = <4, 3, vmcnt(0) per phase> (the field fallback without a wait table).  All
    // three are bit-identical (tests/test_determinism_gpu.py, tests/test_flags_gpu.py).
    constexpr int SMEM4 = 3 * SLAB + 4 * XBUF, SMEM4C = 3 * SLAB + 4 * 2 * XBUF;
    const int v = far_get_tuning(11);
#ifdef FAR_RING_EXP
    // experiment build (tools/ring_ab.py): v & 4 = the barrier WITHOUT its lgkmcnt(0) -- the rounds-3..5 code that raced
    switch (v & 7) {
        case 4: return launch(k_attn128<4, 3, true, SPLIT, false>, 4, SMEM4C);
        case 5: return launch(k_attn128<WAVES, RING, true, SPLIT, false>, WAVES, SMEM_NEW);
        case 6: return launch(k_attn128<4, 3, false, SPLIT, false>, 4, SMEM4);
        default: break;
    }
#endif
    switch (v & 3) {
        case 1: return launch(k_attn128<WAVES, RING, true, SPLIT>, WAVES, SMEM_NEW);
        case 2: return launch(k_attn128<4, 3, false, SPLIT>, 4, SMEM4);
        default: return launch(k_attn128<4, 3, true, SPLIT>, 4, SMEM4C);
    }
}

extern "C" {

int far_attn_block_f16s(const float* x, const float* src, const void* packed, long nwin, int L, int S, int d_model, int heads,
                        float scale_k, float scale_v, float scale_q, float scale_m, float attn_eps, const float* gamma,
                        const float* beta, float ln_eps, float* out, int* overflow, hipStream_t stream) {
    return attn_block_launch<true>(x, src, packed, nwin, L, S, d_model, heads, scale_k, scale_v, scale_q, scale_m, attn_eps, gamma, beta,
                                   ln_eps, out, overflow, stream);
}

// The same block on plain fp16 operands (one MFMA per product; the 16-bit-operand class, LoFTR.set_precision('fp16')): same
// arguments, same packed image (its lo planes are loaded and not used), same activation-range flag.
int far_attn_block_f16(const float* x, const float* src, const void* packed, long nwin, int L, int S, int d_model, int heads,
                       float scale_k, float scale_v, float scale_q, float scale_m, float attn_eps, const float* gamma,
                       const float* beta, float ln_eps, float* out, int* overflow, hipStream_t stream) {
    return attn_block_launch<false>(x, src, packed, nwin, L, S, d_model, heads, scale_k, scale_v, scale_q, scale_m, attn_eps, gamma, beta,
                                    ln_eps, out, overflow, stream);
}

}  // extern "C"
