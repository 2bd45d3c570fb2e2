// K9 for few rows: a Linear layer over the tokens of one pair (a few thousand rows), where K9's workgroup tile -- 64+ rows x
// 128+ columns walked through 16 barrier-separated phases -- leaves most CUs idle and is a 16 us latency chain per launch
// (tools/k9_timing.py linear_small).  Same operator, same packed weight image, same arithmetic in the
// same order per output (bit-identical to K9): y = act(x W^T * scale + shift (+ res)), split-fp16 operands, fp32 accumulate.
//
// Replaces the nn.Linear layers of mp3d_loftr/src/loftr/loftr_module/transformer.py:25-35 at training / small-batch sizes.
//
// Shape of the work: a wave owns 32 rows x 64 columns (NCT = 2 column tiles; up to 256 input channels) for the WHOLE K.
// The four waves of a workgroup take four row tiles and share one column block, whose weights (all of K: <= 64 KiB) are
// brought into LDS by ONE round of LDS-DMA, waited for once, behind ONE barrier -- no ring, no phases.  The rows are read
// straight from global memory as the MFMA A operand (8 consecutive channels per lane), 128 channels per register buffer, two
// buffers in flight, split into (hi, lo) in registers.  The epilogue stores each accumulator register as 32 consecutive
// channels of a row (128-byte segments).
#include "linear_small.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int QK = 128;                    // input channels per register buffer (8 k-steps)

struct ABuf {
    float4 v[16];                          // k-step s: v[2 s], v[2 s + 1] = channels 16 s + 8 h + 0..7 of this lane's row
};

__device__ __forceinline__ void load_quarter(ABuf& b, const float* __restrict__ xrow, bool live, int q, int Cin, int h) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int ch = QK * q + 16 * s + 8 * h;
        const bool ok = live && ch < Cin;                       // Cin % 32 == 0: a k-step is inside or outside as a whole
        b.v[2 * s] = ok ? *reinterpret_cast<const float4*>(xrow + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        b.v[2 * s + 1] = ok ? *reinterpret_cast<const float4*>(xrow + ch + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int NCT>
__global__ __launch_bounds__(256, 2) void k_lin_small(const LinSmallArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [chunk][k-step][plane][32 NCT rows][32 B]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int co0 = blockIdx.y * (32 * NCT);
    const int by = co0 / p.NT, nn0 = co0 - by * p.NT;
    const int nchunks = p.Cin >> 5;
    // ---- the column block's weights, all of K: piece (chunk, k-step, plane) = 32 NCT consecutive rows of 32 bytes in the image
    constexpr int PIECE = 32 * NCT * 32;
    {
        const int npieces = nchunks * 4;
        const int per_piece = PIECE / 1024;                                   // wave-requests of 1 KiB per piece
        for (int i = wave; i < npieces * per_piece; i += 4) {
            const int piece = i / per_piece, sub = i - piece * per_piece;
            const int chunk = piece >> 2, ks = (piece >> 1) & 1, plane = piece & 1;
            const size_t row = ((((size_t)chunk * 2 + ks) * p.nblkY + by) * 2 + plane) * p.NT + nn0;
            const unsigned char* src = p.w + row * 32 + sub * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + (size_t)piece * PIECE + sub * 1024), 16, 0, 0);
        }
    }
    // ---- this wave's rows
    const long row = ((long)blockIdx.x * 4 + wave) * 32 + l31;
    const bool live = row < p.rows;
    const float* xrow = p.x + (live ? row : 0) * p.Cin;
    const float act_scale = p.scale_dev ? p.scale_dev[0] : p.act_scale;
    const float out_mul = p.scale_dev ? p.scale_dev[1] : p.out_mul;
    const int nq = (p.Cin + QK - 1) / QK;
    ABuf b0, b1;
    load_quarter(b0, xrow, live, 0, p.Cin, h);
    if (nq > 1) load_quarter(b1, xrow, live, 1, p.Cin, h);
    f32x16 acc[NCT];
#pragma unroll
    for (int nt = 0; nt < NCT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // weights (and the first row buffers) have landed
    __syncthreads();

    // weight row n of a piece: 2 slots of 8 channels, slot ^= (n >> 3) & 1 (K9's conflict-free layout)
    int boff[NCT];
#pragma unroll
    for (int nt = 0; nt < NCT; ++nt) {
        const int n = nn0 + 32 * nt + l31;                                    // row inside the NT block: the swizzle uses it
        boff[nt] = (32 * nt + l31) * 32 + ((h ^ ((n >> 3) & 1)) * 16);
    }
    auto compute = [&](const ABuf& b, int q) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int kstep = 8 * q + s;                                      // global k-step: chunk = kstep >> 1, ks = kstep & 1
            if (16 * kstep >= p.Cin) break;                                   // wave-uniform
            f16x8 ah, al;
            {
                const float xs[8] = {b.v[2 * s].x, b.v[2 * s].y, b.v[2 * s].z, b.v[2 * s].w,
                                     b.v[2 * s + 1].x, b.v[2 * s + 1].y, b.v[2 * s + 1].z, b.v[2 * s + 1].w};
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    f16x2 hh, ll;
                    split2(f32x2{xs[i], xs[i + 1]} * f32x2{act_scale, act_scale}, hh, ll);
                    ah[i] = hh.x; ah[i + 1] = hh.y;
                    al[i] = ll.x; al[i + 1] = ll.y;
                }
            }
            const unsigned char* B = lds + (size_t)(kstep * 2) * PIECE;       // piece (chunk, ks, plane 0); plane 1 follows
#pragma unroll
            for (int nt = 0; nt < NCT; ++nt) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(B + boff[nt]);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(B + PIECE + boff[nt]);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[nt], 0, 0, 0);
            }
        }
    };
    for (int q = 0; q < nq; q += 2) {
        compute(b0, q);
        if (q + 2 < nq) load_quarter(b0, xrow, live, q + 2, p.Cin, h);
        if (q + 1 < nq) {
            compute(b1, q + 1);
            if (q + 3 < nq) load_quarter(b1, xrow, live, q + 3, p.Cin, h);
        }
    }
    // ---- activation-range guard (as K9)
    if (p.overflow) {
        float chk = 0.f;
#pragma unroll
        for (int nt = 0; nt < NCT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk += acc[nt][r];
        const bool bad = !(fabsf(chk) <= FLT_MAX);
        if (__any(bad) && lane == 0) atomicOr(p.overflow, 1);
    }
    // ---- epilogue: register r of lane (l31, h) = output (row r0 + mfma32_row(r, h), column co0 + 32 nt + l31)
    const long r0 = ((long)blockIdx.x * 4 + wave) * 32;
#pragma unroll
    for (int nt = 0; nt < NCT; ++nt) {
        const int co = co0 + 32 * nt + l31;
        if (co >= p.Cout) continue;
        const float sc = p.scale[co] * out_mul, sh = p.shift ? p.shift[co] : 0.f;
        const int plane = co / p.Csub;
        const long base = (long)plane * p.rows * p.Csub + (co - plane * p.Csub);
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long rr = r0 + mfma32_row(r, h);
            rv[r] = (p.res && rr < p.rows) ? p.res[base + rr * p.Csub] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long rr = r0 + mfma32_row(r, h);
            float v = acc[nt][r] * sc + sh + rv[r];
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = v > 0.f ? v : v * p.slope;
            if (rr < p.rows) p.y[base + rr * p.Csub] = v;
        }
    }
}

template <int NCT>
int launch(const LinSmallArgs& a, hipStream_t stream) {
    const int smem = a.Cin * NCT * 128;
    bool cfg_failed = false;
    FAR_ONCE_PER_DEVICE(cfg_failed = hipFuncSetAttribute((const void*)k_lin_small<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess);
    if (cfg_failed) return far_check_launch();
    dim3 grid((unsigned)((a.rows + 127) / 128), (unsigned)((a.Cout + 32 * NCT - 1) / (32 * NCT)));
    hipLaunchKernelGGL(k_lin_small<NCT>, grid, dim3(256), smem, stream, a);
    return far_check_launch();
}

}  // namespace

bool far_linear_small_covers(long rows, int Cin, int Cout) {
    // whole k-steps and chunks; at most 256 input channels: every column block re-reads its rows, and at K = 512 (32-column
    // blocks) that costs more than K9's staging saves (19200 x 512 -> 512: 113 us against K9's 52); at most 320 workgroups:
    // beyond that K9's staged rows win (19200 x 256 -> 256: 29 us here, 28 us there; 4800 rows: 13.4 against 17.6)
    return rows > 0 && Cin >= 32 && Cin <= 256 && (Cin & 31) == 0 && Cout > 0 && ((rows + 127) / 128) * ((Cout + 63) / 64) <= 320;
}

int far_linear_small_launch(const LinSmallArgs& a, hipStream_t stream) { return launch<2>(a, stream); }
