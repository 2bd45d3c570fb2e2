// K10: the backbone stem -- 7x7 stride-2 convolution of a 1-channel image + BatchNorm + ReLU -- writing NHWC for K9 / K17.  The bare
// convolution of the training forward runs on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain); the
// inference form since round 4 on the f16 matrix cores with split-precision operands (k_stem_f16s below).
//
// Replaces mp3d_loftr/src/loftr/backbone/resnet_fpn.py:60-62   self.conv1 (7x7, stride 2, pad 3, no bias), bn1, relu
//                                                     :103      x0 = self.relu(self.bn1(self.conv1(x)))
// A GEMM with M = output pixels, K = 49 taps (padded to 50), N = Cout: one workgroup = 4 waves = 4 output rows x 32
// columns; the (2*4+5) x (2*32+5) input patch and the [50][Cout] weights live in LDS; the A operand is gathered from
// the patch per tap (stride-2 ds_read_b32: conflict free), so no im2col is ever materialised.
#include "common.h"

namespace {

constexpr int SR = 4, SC = 32;                 // output rows / columns per workgroup
constexpr int PH = 2 * SR + 5, PW = 2 * SC + 5;
constexpr int KP = 50;                         // taps padded to an even count

constexpr int YB = 15;                         // tiles (of 4 output rows) per workgroup: the [50][Cout] weight image (25 KiB from L2) is staged once for them (1.32 -> 1.15 ms per 64 images; 6: 1.18)

template <int NTILES, bool PLAIN>              // Cout = 32 * NTILES; PLAIN: the bare convolution (training: BatchNorm follows with batch statistics)
__global__ __launch_bounds__(256) void k_stem(const float* __restrict__ img, const float* __restrict__ w,
                                              const float* __restrict__ scale, const float* __restrict__ shift, int N,
                                              int H, int W, int Ho, int Wo, int tilesX, int tilesY, float* __restrict__ y) {
    constexpr int C = 32 * NTILES;
    __shared__ float patch[PH * PW];
    __shared__ float wl[KP * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int nby = (tilesY + YB - 1) / YB;
    int t = blockIdx.x;
    const int tx = t % tilesX;
    t /= tilesX;
    const int tyb = t % nby, n = t / nby;
    const int ox0 = tx * SC;
    const float* im = img + (size_t)n * H * W;
    // weights: torch layout [Cout][1][7][7] -> LDS [tap][Cout]; the pad tap is zero
    for (int i = tid; i < KP * C; i += 256) {
        const int tap = i / C, co = i - tap * C;
        wl[i] = tap < 49 ? w[co * 49 + tap] : 0.f;
    }
    float sc[NTILES], sh[NTILES];
#pragma unroll
    for (int nt = 0; nt < NTILES; ++nt) {
        sc[nt] = PLAIN ? 1.f : scale[32 * nt + l31];
        sh[nt] = PLAIN ? 0.f : shift[32 * nt + l31];
    }
    for (int ty = tyb * YB; ty < min(tilesY, (tyb + 1) * YB); ++ty) {
        const int oy0 = ty * SR;
        __syncthreads();                                       // the previous tile's patch is consumed (and wl is written)
        for (int i = tid; i < PH * PW; i += 256) {
            const int py = i / PW, px = i - py * PW;
            const int iy = 2 * oy0 - 3 + py, ix = 2 * ox0 - 3 + px;
            patch[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? im[(size_t)iy * W + ix] : 0.f;
        }
        __syncthreads();

        f32x16 acc[NTILES];
#pragma unroll
        for (int nt = 0; nt < NTILES; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const int abase = (2 * wave) * PW + 2 * l31;           // patch offset of this lane's output pixel (row wave, col l31)
#pragma unroll 5
        for (int s = 0; s < KP / 2; ++s) {
            const int tap = 2 * s + h;                         // this lane half's k index
            const int dy = tap / 7, dx = tap - 7 * dy;
            const float a = tap < 49 ? patch[abase + dy * PW + dx] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NTILES; ++nt) {
                const float b = wl[tap * C + 32 * nt + l31];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
            }
        }
        const int oy = oy0 + wave;
        if (oy < Ho) {
#pragma unroll
            for (int nt = 0; nt < NTILES; ++nt) {
                const int co = 32 * nt + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ox = ox0 + mfma32_row(r, h);
                    if (ox < Wo)
                        y[(((size_t)n * Ho + oy) * Wo + ox) * C + co] = PLAIN ? acc[nt][r] : fmaxf(acc[nt][r] * sc[nt] + sh[nt], 0.f);
                }
            }
        }
    }
}

// Round 4: the inference form (BatchNorm folded, ReLU) on the f16 matrix cores with split-precision operands, as K9 computes its
// layers: K = 49 taps padded to 64 = four k-steps of v_mfma_f32_32x32x16_f16, every product as hi.hi + hi.lo + lo.hi with fp32
// accumulation (fp32-grade; the exact-f32 instruction above retires 1/16 of the products per cycle and made the kernel run at the
// SUM of its matrix time and its store time: 1.10 ms per 64 images, of which 0.5 ms matrix work).  Same tiling and patch; the pixel
// values are scaled by 2^4 and the weights by a power of two taken from their maximum (2^13 <= max |w| 2^e < 2^14; every workgroup
// derives the same e) so that the lo parts stay normal fp16 numbers; the weights' fragments are split once per workgroup into LDS
// [k-step][channel tile][plane][lane][8].
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
constexpr float PIX_SCALE = 16.0f;      // as K9's default activation scale: |pixel| <= 4094 (images in 0..1 or 0..255); a darker pixel's lo part may be a
                                        // subnormal fp16 number: absolute error <= 2^-29 of unit scale, below fp32 resolution of the 49-term sum

template <int NTILES>
__global__ __launch_bounds__(256) void k_stem_f16s(const float* __restrict__ img, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift, int N,
                                                   int H, int W, int Ho, int Wo, int tilesX, int tilesY, float* __restrict__ y) {
    constexpr int C = 32 * NTILES;
    __shared__ float patch[PH * PW];
    __shared__ __attribute__((aligned(16))) _Float16 wf[4 * NTILES * 2 * 64 * 8];
    __shared__ float wred[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int nby = (tilesY + YB - 1) / YB;
    int t = blockIdx.x;
    const int tx = t % tilesX;
    t /= tilesX;
    const int tyb = t % nby, n = t / nby;
    const int ox0 = tx * SC;
    const float* im = img + (size_t)n * H * W;
    // max |w| -> power-of-two weight scale (identical in every workgroup: the same data, a max is order-independent)
    float m = 0.f;
    for (int i = tid; i < 49 * C; i += 256) m = fmaxf(m, fabsf(w[i]));
    wred[tid] = m;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (tid < d) wred[tid] = fmaxf(wred[tid], wred[tid + d]);
        __syncthreads();
    }
    int ex = 0;
    (void)frexpf(wred[0], &ex);                                    // max = f 2^ex, f in [0.5, 1)
    const float wmul = wred[0] > 0.f ? ldexpf(1.0f, 14 - ex) : 1.0f;     // max |w| wmul in [2^13, 2^14)
    const float outmul = 1.0f / (wmul * PIX_SCALE);
    // weight fragments: item = (k-step s, tile nt, lane): 8 taps 16 s + 8 (lane >> 5) + e of channel 32 nt + (lane & 31)
    for (int i = tid; i < 4 * NTILES * 64; i += 256) {
        const int ln = i & 63, nt = (i >> 6) % NTILES, ks = i / (64 * NTILES);
        const int co = 32 * nt + (ln & 31);
        f16x8s vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tap = 16 * ks + 8 * (ln >> 5) + e;
            const float v = tap < 49 ? w[co * 49 + tap] * wmul : 0.f;
            const _Float16 hh = (_Float16)v;
            vh[e] = hh; vl[e] = (_Float16)(v - (float)hh);
        }
        _Float16* dst = wf + (size_t)((ks * NTILES + nt) * 2) * 512 + ln * 8;
        *reinterpret_cast<f16x8s*>(dst) = vh;
        *reinterpret_cast<f16x8s*>(dst + 512) = vl;
    }
    float sc[NTILES], sh[NTILES];
#pragma unroll
    for (int nt = 0; nt < NTILES; ++nt) { sc[nt] = scale[32 * nt + l31] * outmul; sh[nt] = shift[32 * nt + l31]; }
    // patch offsets of this lane's 8 taps per k-step (tap >= 49: offset 0, value masked to zero)
    int toff[4][8];
    unsigned tmask = 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tap = 16 * ks + 8 * h + e;
            toff[ks][e] = tap < 49 ? (tap / 7) * PW + tap % 7 : 0;
            if (tap < 49) tmask |= 1u << (8 * ks + e);
        }
    // The patch of tile t + 1 is requested into registers before tile t is computed (round 6): the image loads -- 3.6 KB per tile, a full
    // HBM round trip -- used to sit between two barriers at the head of every tile, where nothing of this workgroup could hide them.
    constexpr int PIT = (PH * PW + 255) / 256;
    float nxt[PIT];
    auto request = [&](int ty) {
        const int oy0 = ty * SR;
#pragma unroll
        for (int j = 0; j < PIT; ++j) {
            const int i = tid + 256 * j;
            const int py = i / PW, px = i - py * PW;
            const int iy = 2 * oy0 - 3 + py, ix = 2 * ox0 - 3 + px;
            nxt[j] = (i < PH * PW && ty < tilesY && iy >= 0 && iy < H && ix >= 0 && ix < W) ? im[(size_t)iy * W + ix] * PIX_SCALE : 0.f;
        }
    };
    const int ty_end = min(tilesY, (tyb + 1) * YB);
    request(tyb * YB);
    for (int ty = tyb * YB; ty < ty_end; ++ty) {
        const int oy0 = ty * SR;
        __syncthreads();                                       // the previous tile's patch is consumed (and wf is written)
#pragma unroll
        for (int j = 0; j < PIT; ++j)
            if (tid + 256 * j < PH * PW) patch[tid + 256 * j] = nxt[j];
        __syncthreads();
        if (ty + 1 < ty_end) request(ty + 1);                  // in flight under this tile's gathers, MFMAs and stores
        f32x16 acc[NTILES];
#pragma unroll
        for (int nt = 0; nt < NTILES; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const int abase = (2 * wave) * PW + 2 * l31;           // patch offset of this lane's output pixel (row wave, col l31)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8s ah, al;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float v0 = ((tmask >> (8 * ks + e)) & 1u) ? patch[abase + toff[ks][e]] : 0.f;
                const float v1 = ((tmask >> (8 * ks + e + 1)) & 1u) ? patch[abase + toff[ks][e + 1]] : 0.f;
                f16x2 hh, ll;
                split2(f32x2{v0, v1}, hh, ll);
                ah[e] = hh.x; ah[e + 1] = hh.y; al[e] = ll.x; al[e + 1] = ll.y;
            }
#pragma unroll
            for (int nt = 0; nt < NTILES; ++nt) {
                const _Float16* src = wf + (size_t)((ks * NTILES + nt) * 2) * 512 + lane * 8;
                const f16x8s bh = *reinterpret_cast<const f16x8s*>(src), bl = *reinterpret_cast<const f16x8s*>(src + 512);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[nt], 0, 0, 0);
            }
        }
        const int oy = oy0 + wave;
        if (oy < Ho) {
#pragma unroll
            for (int nt = 0; nt < NTILES; ++nt) {
                const int co = 32 * nt + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ox = ox0 + mfma32_row(r, h);
                    // ReLU that keeps NaN (an inf / NaN pixel turns into NaN in the split; fmaxf(NaN, 0) = 0 would hide it from the
                    // activation-range flag of the layers behind: a non-finite image must stay loud)
                    const float v = acc[nt][r] * sc[nt] + sh[nt];
                    if (ox < Wo) y[(((size_t)n * Ho + oy) * Wo + ox) * C + co] = v < 0.f ? 0.f : v;
                }
            }
        }
    }
}

// Weight gradient of the stem (training): dW[co][ky][kx] = sum_{n, oy, ox} dy[n][oy][ox][co] * img[n][2 oy + ky - 3][2 ox + kx - 3],
// the same exact-f32 MFMA with the roles turned: M = Cout, N = 49 taps (two 32-wide tiles, the tail zero), K = pixels.  A lane
// feeds A[co][pixel] = dy[pixel][co] straight from memory (NHWC: 32 consecutive channels per half-wave) and B[pixel][tap]
// from the same LDS patch the forward kernel gathers from.  A wave owns one output row of each tile and walks a run of tiles;
// its [Cout][64] partial goes to a slab and k_stem_wgrad_reduce adds the slabs in index order (deterministic).
template <int NTILES>
__global__ __launch_bounds__(256) void k_stem_wgrad(const float* __restrict__ img, const float* __restrict__ dy, int N, int H, int W,
                                                    int Ho, int Wo, int tilesX, int tilesY, int tiles_per_wg, long ntiles,
                                                    float* __restrict__ part) {
    constexpr int C = 32 * NTILES;
    __shared__ float patch[PH * PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    int toff[2];
    bool tval[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int tap = 32 * tt + l31;
        tval[tt] = tap < 49;
        toff[tt] = tval[tt] ? (tap / 7) * PW + tap % 7 : 0;
    }
    f32x16 acc[NTILES][2];
#pragma unroll
    for (int ct = 0; ct < NTILES; ++ct)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][tt][r] = 0.f;
    const long t0 = (long)blockIdx.x * tiles_per_wg, t1 = t0 + tiles_per_wg < ntiles ? t0 + tiles_per_wg : ntiles;
    for (long t = t0; t < t1; ++t) {
        const int tx = (int)(t % tilesX), ty = (int)((t / tilesX) % tilesY), n = (int)(t / ((long)tilesX * tilesY));
        const int oy0 = ty * SR, ox0 = tx * SC;
        const float* im = img + (size_t)n * H * W;
        __syncthreads();
        for (int i = tid; i < PH * PW; i += 256) {
            const int py = i / PW, px = i - py * PW;
            const int iy = 2 * oy0 - 3 + py, ix = 2 * ox0 - 3 + px;
            patch[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? im[(size_t)iy * W + ix] : 0.f;
        }
        __syncthreads();
        const int oy = oy0 + wave;
        if (oy >= Ho) continue;                                   // wave-uniform; the barriers above are outside
        const float* drow = dy + ((size_t)n * Ho + oy) * Wo * C;
#pragma unroll 4
        for (int c = 0; c < SC; c += 2) {
            const int ox = ox0 + c + h;
            const bool live = ox < Wo;
            float b[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) b[tt] = tval[tt] ? patch[(2 * wave) * PW + 2 * (c + h) + toff[tt]] : 0.f;
#pragma unroll
            for (int ct = 0; ct < NTILES; ++ct) {
                const float a = live ? drow[(size_t)ox * C + 32 * ct + l31] : 0.f;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc[ct][tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[tt], acc[ct][tt], 0, 0, 0);
            }
        }
    }
    // slab of this wave: [Cout][64]; accumulator register r of lane (l31, h) = D[co = 32 ct + row(r, h)][tap = 32 tt + l31]
    float* const slab = part + ((size_t)blockIdx.x * 4 + wave) * C * 64;
#pragma unroll
    for (int ct = 0; ct < NTILES; ++ct)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(32 * ct + mfma32_row(r, h)) * 64 + 32 * tt + l31] = acc[ct][tt][r];
}

// dw[co][tap] = sum over the slabs, in a fixed order: sixteen strided sums per element by sixteen threads, added in group order
__global__ __launch_bounds__(1024) void k_stem_wgrad_reduce(const float* __restrict__ part, int slabs, int C, float* __restrict__ dw) {
    __shared__ float q[16][64];
    const int l = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + l;                               // over [C][64]
    float s = 0.f;
    if (i < C * 64)
        for (int k = grp; k < slabs; k += 16) s += part[(size_t)k * C * 64 + i];
    q[grp][l] = s;
    __syncthreads();
    if (grp == 0 && i < C * 64) {
        float t = q[0][l];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += q[g][l];
        const int co = i >> 6, tap = i & 63;
        if (tap < 49) dw[co * 49 + tap] = t;
    }
}

int stem_wgrad_wgs(int N, int Ho, int Wo, int* tiles_per_wg, long* ntiles) {
    const int tilesX = (Wo + SC - 1) / SC, tilesY = (Ho + SR - 1) / SR;
    *ntiles = (long)N * tilesX * tilesY;
    long per = (*ntiles + 255) / 256;                            // <= 256 workgroups = 1024 slabs of [Cout][64]
    if (per < 1) per = 1;
    *tiles_per_wg = (int)per;
    return (int)((*ntiles + per - 1) / per);
}

}  // namespace

extern "C" {

// y[n][oy][ox][co] = relu(scale[co] * sum_{ky,kx} img[n][2 oy + ky - 3][2 ox + kx - 3] * w[co][ky][kx] + shift[co])
// img [N][H][W] fp32 (one channel), w [Cout][7][7], y [N][Ho][Wo][Cout] NHWC with Ho = (H + 1) / 2, Wo = (W + 1) / 2;
// Cout = 64 or 128.  scale = shift = NULL: the bare convolution, no BatchNorm fold and no ReLU (the training forward).
int far_stem7x7_nhwc_f32(const float* img, const float* w, const float* scale, const float* shift, int N, int H, int W,
                         int Cout, float* y, hipStream_t stream) {
    far_clear_errors();
    if (N == 0) return FAR_OK;
    if (!img || !w || (!scale) != (!shift) || !y || N < 0 || H <= 0 || W <= 0 || (Cout != 64 && Cout != 128)) return FAR_EINVAL;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int tilesX = (Wo + SC - 1) / SC, tilesY = (Ho + SR - 1) / SR;
    const long nb = (long)N * tilesX * ((tilesY + YB - 1) / YB);
    if (nb > 0x7fffffffL) return FAR_EINVAL;
    const bool plain = !scale && !shift;
    if (Cout == 128) {
        if (plain) hipLaunchKernelGGL((k_stem<4, true>), dim3((unsigned)nb), dim3(256), 0, stream, img, w, scale, shift, N, H, W, Ho, Wo, tilesX, tilesY, y);
        else if (far_get_tuning(12) == 0) hipLaunchKernelGGL((k_stem_f16s<4>), dim3((unsigned)nb), dim3(256), 0, stream, img, w, scale, shift, N, H, W, Ho, Wo, tilesX, tilesY, y);
        else hipLaunchKernelGGL((k_stem<4, false>), dim3((unsigned)nb), dim3(256), 0, stream, img, w, scale, shift, N, H, W, Ho, Wo, tilesX, tilesY, y);
    } else {
        if (plain) hipLaunchKernelGGL((k_stem<2, true>), dim3((unsigned)nb), dim3(256), 0, stream, img, w, scale, shift, N, H, W, Ho, Wo, tilesX, tilesY, y);
        else if (far_get_tuning(12) == 0) hipLaunchKernelGGL((k_stem_f16s<2>), dim3((unsigned)nb), dim3(256), 0, stream, img, w, scale, shift, N, H, W, Ho, Wo, tilesX, tilesY, y);
        else hipLaunchKernelGGL((k_stem<2, false>), dim3((unsigned)nb), dim3(256), 0, stream, img, w, scale, shift, N, H, W, Ho, Wo, tilesX, tilesY, y);
    }
    return far_check_launch();
}

// Bytes of device scratch far_stem7x7_wgrad_f32 needs.
long far_stem7x7_wgrad_ws_bytes(int N, int H, int W, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || (Cout != 64 && Cout != 128)) return 0;
    int per; long nt;
    const int wgs = stem_wgrad_wgs(N, (H + 1) / 2, (W + 1) / 2, &per, &nt);
    return (long)wgs * 4 * Cout * 64 * (long)sizeof(float);
}

// dw [Cout][7][7] (overwritten) = sum_{n, oy, ox} dy[n][oy][ox][co] * img[n][2 oy + ky - 3][2 ox + kx - 3]: the weight gradient of
// the stem convolution from the image [N][H][W] and dy [N][Ho][Wo][Cout] (NHWC).  Exact fp32 products, deterministic sum.
int far_stem7x7_wgrad_f32(const float* img, const float* dy, int N, int H, int W, int Cout, void* ws, long ws_bytes, float* dw,
                          hipStream_t stream) {
    far_clear_errors();
    const long need = far_stem7x7_wgrad_ws_bytes(N, H, W, Cout);
    if (!img || !dy || !dw || !ws || need == 0 || ws_bytes < need) return FAR_EINVAL;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int tilesX = (Wo + SC - 1) / SC, tilesY = (Ho + SR - 1) / SR;
    int per; long nt;
    const int wgs = stem_wgrad_wgs(N, Ho, Wo, &per, &nt);
    float* const part = reinterpret_cast<float*>(ws);
    if (Cout == 128)
        hipLaunchKernelGGL(k_stem_wgrad<4>, dim3((unsigned)wgs), dim3(256), 0, stream, img, dy, N, H, W, Ho, Wo, tilesX, tilesY, per, nt, part);
    else
        hipLaunchKernelGGL(k_stem_wgrad<2>, dim3((unsigned)wgs), dim3(256), 0, stream, img, dy, N, H, W, Ho, Wo, tilesX, tilesY, per, nt, part);
    hipLaunchKernelGGL(k_stem_wgrad_reduce, dim3((unsigned)(Cout)), dim3(1024), 0, stream, part, wgs * 4, Cout, dw);
    return far_check_launch();
}

}  // extern "C"
