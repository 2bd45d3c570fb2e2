// K7 / K8: the memory-bound glue of the ResNet-FPN backbone, fused into single HBM passes (inference only).
//
// Replaces, around the vendor convolutions of mp3d_loftr/src/loftr/backbone/resnet_fpn.py:
//   K7  :32-40   y = relu(bn1(conv1(x)));  relu(x + bn2(conv2(y)))      (BasicBlock.forward)
//       :103, :80-91  stem bn+relu, downsample bn, FPN bn + LeakyReLU
//       as ONE pass  y = act(x * scale[c] + shift[c] (+ residual))  with the inference BatchNorm folded into a
//       per-channel scale/shift (scale = gamma / sqrt(var + eps), shift = beta - mean * scale).
//   K8  :110-116  x2_out = layer2_outconv(x2) + interpolate(x3_out, 2x, bilinear, align_corners=True)
//       as ONE pass  out = hi + upsample2x(lo).
// At batch 32 the 1/2-resolution maps are 2.5-3.9 GB each: every avoided pass is ~1 ms.  channels_last (NHWC)
// tensors, 16-byte accesses along the channel axis.
#include "common.h"

namespace {

// x, res, y: [rows = N*H*W][C] (NHWC flattened); C % 4 == 0.  act: 0 none, 1 relu, 2 leaky_relu(slope).
__global__ __launch_bounds__(256) void k_affine_act(const float4* __restrict__ x, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const float4* __restrict__ res,
                                                    long nvec, int cvec, int act, float slope, float4* __restrict__ y) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const int c4 = (int)(i % cvec);
        const float4 s = reinterpret_cast<const float4*>(scale)[c4];
        const float4 t = reinterpret_cast<const float4*>(shift)[c4];
        float4 v = x[i];
        v.x = fmaf(v.x, s.x, t.x); v.y = fmaf(v.y, s.y, t.y); v.z = fmaf(v.z, s.z, t.z); v.w = fmaf(v.w, s.w, t.w);
        if (res) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (act == 2) {
            v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
            v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        }
        y[i] = v;
    }
}

// NCHW variant: x viewed as [N*C][HW], HW % 4 == 0; one scale/shift per row.
__global__ __launch_bounds__(256) void k_affine_act_nchw(const float4* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float4* __restrict__ res,
                                                         long nvec, int hwvec, int C, int act, float slope,
                                                         float4* __restrict__ y) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const int c = (int)((i / hwvec) % C);
        const float s = scale[c], t = shift[c];
        float4 v = x[i];
        v.x = fmaf(v.x, s, t); v.y = fmaf(v.y, s, t); v.z = fmaf(v.z, s, t); v.w = fmaf(v.w, s, t);
        if (res) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (act == 2) {
            v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
            v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        }
        y[i] = v;
    }
}

// NCHW variant of the upsample+add: one thread per 4 consecutive output x of one (n, c, Y) row; W % 4 == 0.
__global__ __launch_bounds__(256) void k_upsample2x_add_nchw(const float* __restrict__ lo, const float4* __restrict__ hi,
                                                             long planes, int h, int w, float4* __restrict__ out) {
    const int H = 2 * h, W = 2 * w, wv = W / 4;
    const long nvec = planes * H * wv;
    const float ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const int xv = (int)(i % wv);
        long p = i / wv;
        const int Y = (int)(p % H);
        const long pl = p / H;
        const float sy = ry * (float)Y;
        const int y0 = (int)sy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly = sy - (float)y0, hy = 1.f - ly;
        const float* r0 = lo + (pl * h + y0) * w;
        const float* r1 = lo + (pl * h + y1) * w;
        float4 v = hi[i];
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sx = rx * (float)(4 * xv + k);
            const int x0 = (int)sx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float lx = sx - (float)x0, hx = 1.f - lx;
            o[k] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
        }
        v.x += o[0]; v.y += o[1]; v.z += o[2]; v.w += o[3];
        out[i] = v;
    }
}

// out[n][Y][X][c] = hi[n][Y][X][c] + bilinear(lo)[n][Y][X][c], output (2h x 2w), align_corners = True:
// source coordinate sy = Y * (h-1)/(2h-1)  (torch upsample_bilinear2d, area_pixel_compute_source_index).
__global__ __launch_bounds__(256) void k_upsample2x_add(const float4* __restrict__ lo, const float4* __restrict__ hi,
                                                        int N, int h, int w, int cvec, float4* __restrict__ out) {
    const int H = 2 * h, W = 2 * w;
    const long nvec = (long)N * H * W * cvec;
    const float ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const int c4 = (int)(i % cvec);
        long p = i / cvec;
        const int X = (int)(p % W); p /= W;
        const int Y = (int)(p % H);
        const int n = (int)(p / H);
        const float sy = ry * (float)Y, sx = rx * (float)X;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float4* base = lo + (long)n * h * w * cvec + c4;
        const float4 a = base[((long)y0 * w + x0) * cvec], b = base[((long)y0 * w + x1) * cvec];
        const float4 c = base[((long)y1 * w + x0) * cvec], d = base[((long)y1 * w + x1) * cvec];
        float4 v = hi[i];
        // torch: hy * (hx * a + lx * b) + ly * (hx * c + lx * d)
        v.x += hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x);
        v.y += hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y);
        v.z += hy * (hx * a.z + lx * b.z) + ly * (hx * c.z + lx * d.z);
        v.w += hy * (hx * a.w + lx * b.w) + ly * (hx * c.w + lx * d.w);
        out[i] = v;
    }
}

// Gradient of the 2x bilinear upsampling with respect to its input, in gather form: one thread per input pixel (x 4 channels,
// channels_last) sums the output-gradient pixels whose footprint contains it, rows then columns in increasing order, with the
// interpolation weights recomputed exactly as the forward computes them.  No atomics: the result is the same bits on every
// run (ATen's upsample_bilinear2d_backward scatters with atomicAdd and is not).
__global__ __launch_bounds__(256) void k_upsample2x_bwd(const float4* __restrict__ dout, int N, int h, int w, int cvec,
                                                        float4* __restrict__ dlo) {
    const int H = 2 * h, W = 2 * w;
    const long nvec = (long)N * h * w * cvec;
    const float ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < nvec; t += stride) {
        const int c4 = (int)(t % cvec);
        long p = t / cvec;
        const int j = (int)(p % w); p /= w;
        const int i = (int)(p % h);
        const int n = (int)(p / h);
        // output rows Y with floor(ry * Y) in {i - 1, i}: Y in [2i - 2, 2i + 3] covers them for every h (ry in [1/3, 1/2))
        const int Ya = max(2 * i - 2, 0), Yb = min(2 * i + 3, H - 1);
        const int Xa = max(2 * j - 2, 0), Xb = min(2 * j + 3, W - 1);
        float wx[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int X = Xa + k;
            const float sx = rx * (float)X;
            const int x0 = (int)sx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float lx = sx - (float)x0, hx = 1.f - lx;
            wx[k] = X <= Xb ? (x0 == j ? hx : 0.f) + (x1 == j ? lx : 0.f) : 0.f;
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* base = dout + (long)n * H * W * cvec + c4;
        for (int Y = Ya; Y <= Yb; ++Y) {
            const float sy = ry * (float)Y;
            const int y0 = (int)sy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const float ly = sy - (float)y0, hy = 1.f - ly;
            const float wy = (y0 == i ? hy : 0.f) + (y1 == i ? ly : 0.f);
            if (wy == 0.f) continue;
            float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (wx[k] == 0.f) continue;
                const float4 g = base[((long)Y * W + Xa + k) * cvec];
                row.x = fmaf(wx[k], g.x, row.x); row.y = fmaf(wx[k], g.y, row.y);
                row.z = fmaf(wx[k], g.z, row.z); row.w = fmaf(wx[k], g.w, row.w);
            }
            acc.x = fmaf(wy, row.x, acc.x); acc.y = fmaf(wy, row.y, acc.y);
            acc.z = fmaf(wy, row.z, acc.z); acc.w = fmaf(wy, row.w, acc.w);
        }
        dlo[t] = acc;
    }
}

inline unsigned grid_for(long nvec) {
    long b = (nvec + 255) / 256;
    return (unsigned)(b < 256L * 16 ? (b > 0 ? b : 1) : 256L * 16);
}

}  // namespace

extern "C" {

// y = act(x * scale[c] + shift[c] (+ res)) for activations [N][C][H][W] (nhwc = 0, HW % 4 == 0) or channels_last
// (nhwc = 1, C % 4 == 0).  act: 0 none, 1 ReLU, 2 LeakyReLU(slope).  y may alias x.
int far_affine_act_f32(const float* x, const float* scale, const float* shift, const float* res, long N, int C,
                       long HW, int nhwc, int act, float slope, float* y, hipStream_t stream) {
    far_clear_errors();
    if (N == 0 || HW == 0) return FAR_OK;
    if (!x || !scale || !shift || !y || N < 0 || HW < 0 || C <= 0 || act < 0 || act > 2) return FAR_EINVAL;
    if (nhwc ? (C & 3) : (HW & 3)) return FAR_EINVAL;
    const long nvec = N * C * HW / 4;
    if (nhwc)
        hipLaunchKernelGGL(k_affine_act, dim3(grid_for(nvec)), dim3(256), 0, stream, (const float4*)x, scale, shift,
                           (const float4*)res, nvec, C / 4, act, slope, (float4*)y);
    else
        hipLaunchKernelGGL(k_affine_act_nchw, dim3(grid_for(nvec)), dim3(256), 0, stream, (const float4*)x, scale, shift,
                           (const float4*)res, nvec, (int)(HW / 4), C, act, slope, (float4*)y);
    return far_check_launch();
}

// out = hi + upsample2x_bilinear_align_corners(lo); lo [N][C][h][w], hi/out [N][C][2h][2w] (nhwc = 0, w % 2 == 0)
// or the channels_last layouts (nhwc = 1, C % 4 == 0).
int far_upsample2x_add_f32(const float* lo, const float* hi, int N, int h, int w, int C, int nhwc, float* out,
                           hipStream_t stream) {
    far_clear_errors();
    if (N == 0) return FAR_OK;
    if (!lo || !hi || !out || N < 0 || h <= 0 || w <= 0 || C <= 0) return FAR_EINVAL;
    if (nhwc ? (C & 3) : (w & 1)) return FAR_EINVAL;
    const long nvec = (long)N * C * h * w;     // output float4 count = N*C*(2h)*(2w)/4
    if (nhwc)
        hipLaunchKernelGGL(k_upsample2x_add, dim3(grid_for(nvec)), dim3(256), 0, stream, (const float4*)lo,
                           (const float4*)hi, N, h, w, C / 4, (float4*)out);
    else
        hipLaunchKernelGGL(k_upsample2x_add_nchw, dim3(grid_for(nvec)), dim3(256), 0, stream, lo, (const float4*)hi,
                           (long)N * C, h, w, (float4*)out);
    return far_check_launch();
}

// dlo [N][h][w][C] = gradient of upsample2x_bilinear_align_corners(lo) given dout [N][2h][2w][C] (channels_last, C % 4 == 0);
// fixed summation order, run-to-run bit-identical.  Replaces ATen upsample_bilinear2d_backward (atomics) on the training path.
int far_upsample2x_bwd_f32(const float* dout, int N, int h, int w, int C, float* dlo, hipStream_t stream) {
    far_clear_errors();
    if (N == 0) return FAR_OK;
    if (!dout || !dlo || N < 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3)) return FAR_EINVAL;
    const long nvec = (long)N * h * w * (C / 4);
    hipLaunchKernelGGL(k_upsample2x_bwd, dim3(grid_for(nvec)), dim3(256), 0, stream, (const float4*)dout, N, h, w, C / 4,
                       (float4*)dlo);
    return far_check_launch();
}

}  // extern "C"
