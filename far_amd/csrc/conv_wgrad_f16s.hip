// K16: weight gradient of the backbone's 3x3 / 1x1 convolutions (training), split-fp16 operands on the matrix cores.
//
// Replaces autograd's backward-weights of mp3d_loftr/src/loftr/backbone/resnet_fpn.py:5-12 (conv1x1 / conv3x3, stride 1 or 2,
// 'same' padding, no bias):
//     dW[co][ci][ky][kx] = sum_{n, oy, ox} dy[n][oy][ox][co] * x[n][s oy + ky - pad][s ox + kx - pad][ci]      (NHWC tensors)
// which the training step of round 2 left to the vendor library (MIOpen igemm_wrw, ~6.6 ms of a 60 ms step).
//
// The contraction runs over PIXELS while both tensors are channel-major per pixel, so an MFMA operand (8 consecutive k per
// lane) is 8 consecutive pixels of ONE channel: every lane gathers its own channel with 4-byte loads that are contiguous ACROSS
// the 32 lanes of a half-wave (128-byte segments, the layout NHWC already has) -- no transposition pass, no LDS.  A wave owns a
// 32 co x 32 ci tile of dW for all taps (9 accumulators) and walks a 16-pixel-wide column strip downwards:
//   * one x row (16 + halo pixels) is loaded, scaled, split into fp16 hi + lo ONCE and packed into the three kx-shifted operands;
//   * it is multiplied against the (up to) three dy rows it contributes to (ky = 0, 1, 2), which are kept converted in a
//     rolling register window -- 8 registers per dy row, against 24 for an x row;
//   * each product is hi*hi + hi*lo + lo*hi (fp32-grade, the split of K9), 27 v_mfma_f32_32x32x16_f16 per row step against
//     ~120 VALU conversions: matrix-pipe bound with two waves per SIMD.
// Image borders cost nothing: rows are addressed through a per-row buffer resource, so pixels right of the image (and the
// ragged last strip) read as zero in hardware; the one pixel left of the image is forced out of range; rows above / below are
// skipped.  Lanes of a channel tile that sticks out past Cin / Cout compute garbage in accumulator columns / rows nobody stores.
//
// The (image, strip, row) sequence is cut into ranges; a workgroup is four wave tiles on one range (they share their rows
// through L1); each range writes its partial dW to a workspace slab and k_wgrad_reduce sums the slabs in a fixed order:
// deterministic, no atomics.
#include "common.h"

extern "C" int far_grad_scale_f32(const float* x, long n, float* out2, hipStream_t stream);

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct WsArgs {
    const float* x;          // [N][H][W][Cin]
    const float* dy;         // [N][Ho][Wo][Cout]
    float* part;             // [ranges][taps][Cout][Cin]
    const float* sg_dev;     // device { 2^e, . }: dy is multiplied by sg_dev[0] before the split
    float sx;                // x is multiplied by it before the split (2^act_exp)
    int N, H, W, Ho, Wo, Cin, Cout;
    int strips;              // 16-pixel column strips per image row
    long units, units_per_range;   // a unit = one output row of one strip; ranges are runs of units in (n, strip, oy) order
};

struct Operand {             // 8 pixels of one channel per lane: the A or B operand of v_mfma_f32_32x32x16_f16, split
    f16x8 hi, lo;
};

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}

struct Seg {                 // one segment of a range: output rows [y0, y1) of strip `strip` of image n, and this lane's offsets
    int n, y0, y1;
    unsigned xoff0, aoff0;
    bool left_out;
    float sg;
};

// One dy row of the strip: pixels c0 + 8 kg + 0..7, channel co0 + l31 (raw loads; split by split_dy).
__device__ __forceinline__ void issue_dy(const WsArgs& p, const Seg& c, int oy, float (&v)[8]) {
    const __amdgpu_buffer_rsrc_t r = row_rsrc(p.dy + ((size_t)c.n * p.Ho + oy) * p.Wo * p.Cout, p.Wo * p.Cout * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = buf_load(r, c.aoff0 + (unsigned)(i * p.Cout * 4));
}

__device__ __forceinline__ void split_dy(const float (&v)[8], float sg, Operand& A) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {                       // two pixels per packed conversion (common.h: split2)
        f16x2 h, l;
        split2(f32x2{v[i], v[i + 1]} * f32x2{sg, sg}, h, l);
        A.hi[i] = h.x; A.hi[i + 1] = h.y;
        A.lo[i] = l.x; A.lo[i + 1] = l.y;
    }
}

// One x row: pixels ixb + 0..L-1 (ixb = ST (c0 + 8 kg) - pad), channel ci0 + l31.
template <int KS, int ST>
__device__ __forceinline__ void issue_x(const WsArgs& p, const Seg& c, int iy, float (&v)[ST * 7 + KS]) {
    constexpr int L = ST * 7 + KS;
    const __amdgpu_buffer_rsrc_t r = row_rsrc(p.x + ((size_t)c.n * p.H + iy) * p.W * p.Cin, p.W * p.Cin * 4);
#pragma unroll
    for (int i = 0; i < L; ++i) {
        unsigned off = c.xoff0 + (unsigned)(i * p.Cin * 4);
        if (KS == 3 && i == 0) off = c.left_out ? 0x80000000u : off;     // the pixel left of the image: out of range -> 0
        v[i] = buf_load(r, off);
    }
}

// ... -> the KS kx-shifted split operands: output pixel q of the lane's 8 reads x pixel ST q + kx.
template <int KS, int ST>
__device__ __forceinline__ void split_x(const float (&v)[ST * 7 + KS], float sx, Operand (&B)[KS]) {
    // one packed conversion per operand register (two pixels); identical pairs of different windows (kx = 0 / 2 at stride 1)
    // are the same expression and are computed once
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            f16x2 h, l;
            split2(f32x2{v[ST * q + kx], v[ST * (q + 1) + kx]} * f32x2{sx, sx}, h, l);
            B[kx].hi[q] = h.x; B[kx].hi[q + 1] = h.y;
            B[kx].lo[q] = l.x; B[kx].lo[q + 1] = l.y;
        }
}

// acc[kx] += dy row (A) x the kx-shifted x row (B): hi*hi + hi*lo + lo*hi, the three products of one accumulator KS MFMAs apart.
template <int KS>
__device__ __forceinline__ void mac_row(f32x16* acc, const Operand& A, const Operand (&B)[KS]) {
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi, B[kx].hi, acc[kx], 0, 0, 0);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi, B[kx].lo, acc[kx], 0, 0, 0);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.lo, B[kx].hi, acc[kx], 0, 0, 0);
}

// ---- 3x3 stride 1.  Step j handles input row iy = y0 - 1 + j (j = 3 m + S), which contributes to output rows iy + 1 (ky = 0),
// iy (ky = 1) and iy - 1 (ky = 2); the dy window slot of output row oy is (oy - y0 + 1) % 3.  The raw rows of step j + 1 are
// requested before step j's MFMAs are issued, so their latency is covered by 27 MFMAs.
struct Raw1 {
    float a[8], x[10];
};

__device__ __forceinline__ void prefetch_s1(const WsArgs& p, const Seg& c, int j, Raw1& raw) {
    const int iy = c.y0 - 1 + j;
    if (iy + 1 < c.y1) issue_dy(p, c, iy + 1, raw.a);
    if (iy >= 0 && iy < p.H) issue_x<3, 1>(p, c, iy, raw.x);
}

template <int S>
__device__ __forceinline__ void step_s1(const WsArgs& p, const Seg& c, int j, int nsteps, Raw1& raw, Operand (&A)[3], f32x16 (&acc)[9]) {
    const int iy = c.y0 - 1 + j;
    const bool row = iy >= 0 && iy < p.H;
    Operand B[3];
    if (iy + 1 < c.y1) split_dy(raw.a, c.sg, A[(S + 1) % 3]);
    if (row) split_x<3, 1>(raw.x, p.sx, B);
    if (j + 1 < nsteps) prefetch_s1(p, c, j + 1, raw);
    if (!row) return;
    if (iy + 1 < c.y1) mac_row<3>(acc + 0, A[(S + 1) % 3], B);
    if (iy >= c.y0 && iy < c.y1) mac_row<3>(acc + 3, A[S % 3], B);
    if (iy - 1 >= c.y0) mac_row<3>(acc + 6, A[(S + 2) % 3], B);
}

// ---- 3x3 stride 2, output row m (slot S = (m - y0) & 1): input row 2m - 1 is ky = 0 of row m and ky = 2 of row m - 1; row 2m is
// ky = 1 of row m.  m runs to y1 inclusive (the ky = 2 row of the last output row).
template <int S>
__device__ __forceinline__ void step_s2(const WsArgs& p, const Seg& c, int m, Operand (&A)[2], f32x16 (&acc)[9]) {
    const bool cur = m < c.y1, prev = m > c.y0;
    float va[8], vx[17];
    Operand B[3];
    const int io = 2 * m - 1;
    const bool odd = io >= 0 && io < p.H;
    if (cur) issue_dy(p, c, m, va);
    if (odd) issue_x<3, 2>(p, c, io, vx);
    if (cur) split_dy(va, c.sg, A[S]);
    if (odd) {
        split_x<3, 2>(vx, p.sx, B);
        if (cur) issue_x<3, 2>(p, c, 2 * m, vx);           // the even row travels under the odd row's MFMAs
        if (cur) mac_row<3>(acc + 0, A[S], B);
        if (prev) mac_row<3>(acc + 6, A[S ^ 1], B);
    } else if (cur) {
        issue_x<3, 2>(p, c, 2 * m, vx);
    }
    if (cur) {                                             // 2m <= H - 1 for every m < Ho
        split_x<3, 2>(vx, p.sx, B);
        mac_row<3>(acc + 3, A[S], B);
    }
}

template <int KS, int ST>
__global__ __launch_bounds__(256, (KS == 3 && ST == 2) ? 1 : 2) void k_wgrad_s(const WsArgs p) {
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    // wave tiles in (ci tile, co tile) order, four consecutive ones per workgroup: they mostly share the x rows through L1
    const int nco = (p.Cout + 31) >> 5, nci = (p.Cin + 31) >> 5;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= nco * nci) return;                                         // no barriers in this kernel
    const int co0 = 32 * (tile % nco), ci0 = 32 * (tile / nco);
    const float sg = p.sg_dev[0];

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    long u = (long)blockIdx.y * p.units_per_range;
    const long u1 = u + p.units_per_range < p.units ? u + p.units_per_range : p.units;
    while (u < u1) {
        const int col = (int)(u / p.Ho), strip = col % p.strips;
        Seg c;
        c.n = col / p.strips;
        c.y0 = (int)(u - (long)col * p.Ho);
        c.y1 = (long)c.y0 + (u1 - u) < p.Ho ? c.y0 + (int)(u1 - u) : p.Ho;
        u += c.y1 - c.y0;
        const int c0 = strip * 16;
        const int ixb = ST * (c0 + 8 * kg) - PAD;
        c.xoff0 = (unsigned)((ixb * p.Cin + ci0 + l31) * 4);              // wraps for ixb = -1; pixel i >= 1 wraps back
        c.aoff0 = (unsigned)(((c0 + 8 * kg) * p.Cout + co0 + l31) * 4);
        c.left_out = ixb < 0;
        c.sg = sg;
        if constexpr (KS == 3 && ST == 1) {
            Operand A[3];
            Raw1 raw;
            const int nsteps = c.y1 - c.y0 + 2;
            prefetch_s1(p, c, 0, raw);
            for (int j = 0; j < nsteps; j += 3) {
                step_s1<0>(p, c, j, nsteps, raw, A, acc);
                if (j + 1 < nsteps) step_s1<1>(p, c, j + 1, nsteps, raw, A, acc);
                if (j + 2 < nsteps) step_s1<2>(p, c, j + 2, nsteps, raw, A, acc);
            }
        } else if constexpr (KS == 3) {
            Operand A[2];
            for (int m = c.y0; m <= c.y1; m += 2) {
                step_s2<0>(p, c, m, A, acc);
                if (m + 1 <= c.y1) step_s2<1>(p, c, m + 1, A, acc);
            }
        } else {
            // 1x1: three MFMAs per row are no cover for a load, so rows go four at a time -- the next four are requested
            // before the current four are multiplied
            constexpr int CH = 4;
            float va[CH][8], vx[CH][ST * 7 + 1];
#pragma unroll
            for (int r = 0; r < CH; ++r)
                if (c.y0 + r < c.y1) {
                    issue_dy(p, c, c.y0 + r, va[r]);
                    issue_x<1, ST>(p, c, ST * (c.y0 + r), vx[r]);
                }
            for (int oy = c.y0; oy < c.y1; oy += CH) {
                Operand A[CH], B[CH][1];
#pragma unroll
                for (int r = 0; r < CH; ++r)
                    if (oy + r < c.y1) {
                        split_dy(va[r], sg, A[r]);
                        split_x<1, ST>(vx[r], p.sx, B[r]);
                    }
#pragma unroll
                for (int r = 0; r < CH; ++r)
                    if (oy + CH + r < c.y1) {
                        issue_dy(p, c, oy + CH + r, va[r]);
                        issue_x<1, ST>(p, c, ST * (oy + CH + r), vx[r]);
                    }
#pragma unroll
                for (int r = 0; r < CH; ++r)
                    if (oy + r < c.y1) mac_row<1>(acc, A[r], B[r]);
            }
        }
    }

    // partial dW of this range: accumulator register r of lane (l31, kg) = D[co = row(r, kg)][ci = l31]
    const int ci = ci0 + l31;
    if (ci < p.Cin) {
        float* const slab = p.part + (size_t)blockIdx.y * TAPS * p.Cout * p.Cin;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + mfma32_row(r, kg);
                if (co < p.Cout) slab[((size_t)t * p.Cout + co) * p.Cin + ci] = acc[t][r];
            }
    }
}

// dw[co][ci][t] = (sum over slabs of part[slab][t][co][ci]) / (sx sg), slabs in index order; overflow |= a non-finite sum.
__global__ void k_wgrad_reduce(const float* __restrict__ part, int slabs, int taps, long cc, const float* __restrict__ sg_dev, float inv_sx,
                               float* __restrict__ dw, int* __restrict__ overflow) {
    const long total = cc * taps;
    const float inv = inv_sx / sg_dev[0];
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < slabs; ++k) s += part[(size_t)k * total + i];
        const long t = i / cc, e = i - t * cc;
        bad |= !(fabsf(s) <= FLT_MAX);
        dw[e * taps + t] = s * inv;
    }
    if (overflow && bad) atomicOr(overflow, 1);
}

struct Plan {
    int strips, wgs_per_range;
    long units, units_per_range, slabs;
};

// Ranges: the chip holds 512 workgroups (2 per CU, one wave of each per SIMD).  More ranges spread the rows over more waves but
// every range costs one slab of partial sums (written, then read by the reduction).  The candidates fill 1/2, 1, 2, 3 or 4
// rounds of workgroup slots; the cheapest by a two-term model (row steps x the measured step time + slab traffic) is taken.
Plan plan_for(int N, int Ho, int Wo, int Cin, int Cout, int taps, int stride) {
    Plan pl;
    pl.strips = (Wo + 15) / 16;
    const long tiles = (long)((Cout + 31) / 32) * ((Cin + 31) / 32);
    pl.wgs_per_range = (int)((tiles + 3) / 4);
    pl.units = (long)N * pl.strips * Ho;
    const double slab_us = 2.0 * taps * Cout * (double)Cin * 4.0 / 4.0e6;          // write + read at ~4 TB/s
    const double step_us = taps == 9 ? 0.45 : 0.25;                                 // one wave's row step (1x1: a quarter of a four-row load round trip)
    double best = 1e30;
    long best_r = 1;
    const int resident = taps == 9 && stride == 2 ? 256 : 512;                      // the stride-2 kernel needs a SIMD's whole register file
    const int slots[5] = {256, 512, 1024, 1536, 2048};
    for (int k = 0; k < 5; ++k) {
        if (slots[k] % resident && slots[k] > resident) continue;
        long r = slots[k] / pl.wgs_per_range;
        if (r < 1) r = 1;
        if (r > pl.units / 4) r = pl.units / 4 > 0 ? pl.units / 4 : 1;
        const long per = (pl.units + r - 1) / r;
        const long wgs = r * pl.wgs_per_range;
        // 3x3: matrix-pipe bound -- two resident workgroups per CU share the pipe (x 2 per round of 512); one per CU hides less
        // latency (x 1.3).  1x1: three MFMAs per row, a chain of load round trips -- resident workgroups do not slow each other.
        const double rounds = taps == 1 ? (double)((wgs + 511) / 512)
                            : wgs <= 256 ? 1.3 : (resident == 512 ? 2.0 * (double)((wgs + 511) / 512) : 1.3 * (double)((wgs + 255) / 256));
        const double cost = (per + 4) * step_us * rounds + r * slab_us;
        if (cost < best) { best = cost; best_r = r; }
    }
    pl.units_per_range = (pl.units + best_r - 1) / best_r;
    pl.slabs = (pl.units + pl.units_per_range - 1) / pl.units_per_range;
    return pl;
}

template <int KS, int ST>
void launch_ws(const WsArgs& a, const Plan& pl, hipStream_t stream) {
    hipLaunchKernelGGL((k_wgrad_s<KS, ST>), dim3((unsigned)pl.wgs_per_range, (unsigned)pl.slabs), dim3(256), 0, stream, a);
}

}  // namespace

extern "C" {

// Bytes of workspace far_conv_wgrad_f16s needs for this shape (0 for a shape it does not cover).
long far_conv_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int ksize, int stride) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3) || stride < 1 || stride > 2) return 0;
    const Plan pl = plan_for(N, (H - 1) / stride + 1, (W - 1) / stride + 1, Cin, Cout, ksize * ksize, stride);
    return 256 + pl.slabs * (long)ksize * ksize * Cout * Cin * (long)sizeof(float);
}

// dw [Cout][Cin][ksize][ksize] (torch layout, fp32, overwritten) = the weight gradient of y = conv(x, w) ('same' zero padding
// ksize / 2, stride 1 or 2, no bias) from x [N][H][W][Cin] and dy [N][Ho][Wo][Cout] (NHWC fp32 contiguous,
// Ho = (H - 1) / stride + 1).  x is multiplied by 2^act_exp before the fp16 split (as in far_conv_nhwc_f32: |x| up to
// 65504 / 2^act_exp survive), dy by dy_scale_dev[0] (two device floats of far_grad_scale_f32; NULL: computed here).
// ws: far_conv_wgrad_ws_bytes() bytes of device scratch.  overflow: device int OR-ed with 1 when a sum came out non-finite
// (an operand beyond the split's range), or NULL.  Deterministic: no atomics on dw.
int far_conv_wgrad_f16s(const float* x, const float* dy, int N, int H, int W, int Cin, int Cout, int ksize, int stride, int act_exp,
                        const float* dy_scale_dev, void* ws, long ws_bytes, float* dw, int* overflow, hipStream_t stream) {
    far_clear_errors();
    const long need = far_conv_wgrad_ws_bytes(N, H, W, Cin, Cout, ksize, stride);
    if (!x || !dy || !dw || !ws || need == 0 || ws_bytes < need || act_exp < -24 || act_exp > 15 ||
        (long)W * (Cin > Cout ? Cin : Cout) * 4 >= (1L << 31))
        return FAR_EINVAL;
    WsArgs a;
    a.x = x; a.dy = dy; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    const Plan pl = plan_for(N, a.Ho, a.Wo, Cin, Cout, ksize * ksize, stride);
    a.strips = pl.strips; a.units = pl.units; a.units_per_range = pl.units_per_range;
    float* const scale = reinterpret_cast<float*>(ws);
    a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + 256);
    if (!dy_scale_dev) {
        const int rc = far_grad_scale_f32(dy, (long)N * a.Ho * a.Wo * Cout, scale, stream);
        if (rc != FAR_OK) return rc;
        dy_scale_dev = scale;
    }
    a.sg_dev = dy_scale_dev;
    a.sx = ldexpf(1.0f, act_exp);
    if (ksize == 3) {
        if (stride == 1) launch_ws<3, 1>(a, pl, stream); else launch_ws<3, 2>(a, pl, stream);
    } else {
        if (stride == 1) launch_ws<1, 1>(a, pl, stream); else launch_ws<1, 2>(a, pl, stream);
    }
    const long cc = (long)Cout * Cin;
    const int taps = ksize * ksize;
    const long total = cc * taps;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a.part, (int)pl.slabs, taps, cc,
                       dy_scale_dev, ldexpf(1.0f, -act_exp), dw, overflow);
    return far_check_launch();
}

}  // extern "C"
