// K19: BatchNorm2d with BATCH statistics (training mode) for the ResNet-FPN backbone, forward and backward, on channels_last
// (NHWC) fp32 tensors -- deterministic (fixed-order two-stage sums), fused with the activation and the residual add.
//
// Replaces, under autograd, what torch / MIOpen run for
//   mp3d_loftr/src/loftr/backbone/resnet_fpn.py:24-41  (BasicBlock: relu(bn1(conv1 x)); relu(x' + bn2(conv2 .)); downsample bn)
//                                               :60-62, 103 (stem: relu(bn1(conv1 x)))
//                                               :75-91, 108-117 (layer*_outconv2: conv -> bn -> leaky_relu -> conv)
//
// Forward:  mean_c, var_c (biased) over the M = N H W pixels; y = act((x - mean) rstd gamma + beta (+ residual)); running statistics
//           as nn.BatchNorm2d (momentum, unbiased variance).  far_bn_train_stats_f32 writes scale = gamma rstd, shift = beta - mean scale
//           (the normalisation then IS K7: far_affine_act_f32) and keeps (mean, rstd) for the backward.
// Backward: g = dy act'(y);  dbeta = sum g;  dgamma = sum g xhat, xhat = (x - mean) rstd;
//           dx = gamma rstd (g - dbeta / M - xhat dgamma / M);  the residual's gradient is g.
// Sums: every workgroup adds a contiguous range of pixels per channel in fp32 around a per-range shift (the range's first pixel:
// no cancellation between a large mean and a small variance), the ranges are combined in float64 in a fixed order (four quarter sums per channel, then their sum).
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;

// partial sums over pixel range [p0, p1) of part `blockIdx.x`: MODE 0: (x - s), (x - s)^2 with s = x[p0][c];
// MODE 1: g, g xhat with g = dy act'(y).  out[part][0 | 1][C] (+ out[part][2][C] = s in MODE 0).
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void k_bn_partial(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                           const float4* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, long M, int cvec, long per, int act, float slope,
                                                           float* __restrict__ out) {
    __shared__ float4 red[2][BN_THREADS];
    const int part = blockIdx.x, C = 4 * cvec;
    const long p0 = (long)part * per, p1 = p0 + per < M ? p0 + per : M;
    const int rows = BN_THREADS / cvec;                  // pixel rows handled side by side (cvec <= 256)
    const int c4 = threadIdx.x % cvec, r = threadIdx.x / cvec;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, s = a, mu = a, rs = a;
    if (r < rows && p0 < p1) {
        if (MODE == 0) s = x[p0 * cvec + c4];
        else { mu = reinterpret_cast<const float4*>(mean)[c4]; rs = reinterpret_cast<const float4*>(rstd)[c4]; }
        for (long p = p0 + r; p < p1; p += rows) {
            const float4 v = x[p * cvec + c4];
            if (MODE == 0) {
                const float dx_ = v.x - s.x, dy_ = v.y - s.y, dz_ = v.z - s.z, dw_ = v.w - s.w;
                a.x += dx_; a.y += dy_; a.z += dz_; a.w += dw_;
                b.x = fmaf(dx_, dx_, b.x); b.y = fmaf(dy_, dy_, b.y); b.z = fmaf(dz_, dz_, b.z); b.w = fmaf(dw_, dw_, b.w);
            } else {
                float4 g = dy[p * cvec + c4];
                if (act) {
                    const float4 o = y[p * cvec + c4];
                    const float sl = act == 2 ? slope : 0.f;
                    g.x *= o.x > 0.f ? 1.f : sl; g.y *= o.y > 0.f ? 1.f : sl; g.z *= o.z > 0.f ? 1.f : sl; g.w *= o.w > 0.f ? 1.f : sl;
                }
                a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
                b.x = fmaf(g.x, (v.x - mu.x) * rs.x, b.x); b.y = fmaf(g.y, (v.y - mu.y) * rs.y, b.y);
                b.z = fmaf(g.z, (v.z - mu.z) * rs.z, b.z); b.w = fmaf(g.w, (v.w - mu.w) * rs.w, b.w);
            }
        }
    }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    if (r == 0) {                                        // fixed order over the side-by-side rows
        for (int q = 1; q < rows; ++q) {
            const float4 a2 = red[0][q * cvec + c4], b2 = red[1][q * cvec + c4];
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
            b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        float4* o = reinterpret_cast<float4*>(out + (long)part * 3 * C);
        o[c4] = a; o[cvec + c4] = b;
        if (MODE == 0) o[2 * cvec + c4] = s;
    }
}

// Combine the parts (float64, fixed order): workgroup = 64 channels x 4 threads; thread j of a channel adds the j-th quarter of the
// parts in order, the four quarter sums are added in order 0..3.
template <int MODE>
__device__ __forceinline__ void bn_combine(const float* __restrict__ part, int nparts, long M, long per, int C, int c, int j, double& s0, double& s1) {
    const int q0 = (int)((long)nparts * j / 4), q1 = (int)((long)nparts * (j + 1) / 4);
    double a0 = 0.0, a1 = 0.0;
    if (c < C)
        for (int q = q0; q < q1; ++q) {
            const double a = part[((long)q * 3 + 0) * C + c], b = part[((long)q * 3 + 1) * C + c];
            if (MODE == 0) {                         // sum x = sum (a + n s);  sum x^2 = sum (b + 2 s a + n s^2)
                const long p0 = (long)q * per;
                const double n = (double)((p0 + per < M ? p0 + per : M) - p0);
                if (n <= 0) continue;
                const double s = part[((long)q * 3 + 2) * C + c];
                a0 += a + n * s;
                a1 += b + 2.0 * s * a + n * s * s;
            } else { a0 += a; a1 += b; }
        }
    __shared__ double red[2][4][64];
    red[0][j][threadIdx.x & 63] = a0; red[1][j][threadIdx.x & 63] = a1;
    __syncthreads();
    const int l = threadIdx.x & 63;
    s0 = ((red[0][0][l] + red[0][1][l]) + red[0][2][l]) + red[0][3][l];
    s1 = ((red[1][0][l] + red[1][1][l]) + red[1][2][l]) + red[1][3][l];
}

// write scale / shift / (mean, rstd), update the running statistics
__global__ __launch_bounds__(BN_THREADS) void k_bn_finalize_fwd(const float* __restrict__ part, int nparts, long M, long per, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                float momentum, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, float* __restrict__ scale,
                                                                float* __restrict__ shift, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), j = threadIdx.x >> 6;
    double sx, sxx;
    bn_combine<0>(part, nparts, M, per, C, c, j, sx, sxx);
    if (j == 0 && c < C) {
        const double m = sx / (double)M;
        double var = sxx / (double)M - m * m;
        var = var > 0.0 ? var : 0.0;
        const double rs = 1.0 / sqrt(var + (double)eps);
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        scale[c] = (float)((double)g * rs);
        shift[c] = (float)((double)bt - m * (double)g * rs);
        mean_out[c] = (float)m;
        rstd_out[c] = (float)rs;
        if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (M > 1 ? var * (double)M / (double)(M - 1) : var));
    }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_finalize_bwd(const float* __restrict__ part, int nparts, int C,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), j = threadIdx.x >> 6;
    double sg, sgx;
    bn_combine<1>(part, nparts, 0, 0, C, c, j, sg, sgx);
    if (j == 0 && c < C) { dbeta[c] = (float)sg; dgamma[c] = (float)sgx; }
}

// dx = gamma rstd (g - dbeta / M - xhat dgamma / M); dres = g (optional); g = dy act'(y)
__global__ __launch_bounds__(256) void k_bn_bwd_dx(const float4* __restrict__ x, const float4* __restrict__ dy, const float4* __restrict__ y,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                   const float* __restrict__ dbeta, long nvec, int cvec, float invM, int act, float slope,
                                                   float4* __restrict__ dx, float4* __restrict__ dres) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const int c4 = (int)(i % cvec);
        const float4 mu = reinterpret_cast<const float4*>(mean)[c4], rs = reinterpret_cast<const float4*>(rstd)[c4];
        const float4 dg = reinterpret_cast<const float4*>(dgamma)[c4], db = reinterpret_cast<const float4*>(dbeta)[c4];
        float4 gm = make_float4(1.f, 1.f, 1.f, 1.f);
        if (gamma) gm = reinterpret_cast<const float4*>(gamma)[c4];
        const float4 v = x[i];
        float4 g = dy[i];
        if (act) {
            const float4 o = y[i];
            const float sl = act == 2 ? slope : 0.f;
            g.x *= o.x > 0.f ? 1.f : sl; g.y *= o.y > 0.f ? 1.f : sl; g.z *= o.z > 0.f ? 1.f : sl; g.w *= o.w > 0.f ? 1.f : sl;
        }
        float4 d;
        d.x = gm.x * rs.x * (g.x - db.x * invM - (v.x - mu.x) * rs.x * (dg.x * invM));
        d.y = gm.y * rs.y * (g.y - db.y * invM - (v.y - mu.y) * rs.y * (dg.y * invM));
        d.z = gm.z * rs.z * (g.z - db.z * invM - (v.z - mu.z) * rs.z * (dg.z * invM));
        d.w = gm.w * rs.w * (g.w - db.w * invM - (v.w - mu.w) * rs.w * (dg.w * invM));
        dx[i] = d;
        if (dres) dres[i] = g;
    }
}

// y = act(x scale[c] + shift[c] (+ res)), channels_last (K7's arithmetic; here so that the forward is one library call)
__global__ __launch_bounds__(256) void k_bn_apply(const float4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                  const float4* __restrict__ res, long nvec, int cvec, int act, float slope,
                                                  float4* __restrict__ y) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const int c4 = (int)(i % cvec);
        const float4 s = reinterpret_cast<const float4*>(scale)[c4], t = reinterpret_cast<const float4*>(shift)[c4];
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

inline int bn_parts(long M) {
    long n = (M + 511) / 512;                            // >= 512 pixels per part
    return (int)(n < 1 ? 1 : (n > 512 ? 512 : n));
}

}  // namespace

extern "C" {

// Bytes of the partial-sum scratch of far_bn_train_stats_f32 / far_bn_train_bwd_f32 for M = N H W pixels and C channels.
long far_bn_train_ws_bytes(long M, int C) {
    if (M <= 0 || C <= 0) return 0;
    return (long)bn_parts(M) * 3 * C * (long)sizeof(float);
}

// Batch statistics of x [M][C] (channels_last: M = N H W pixels, C % 4 == 0, C <= 1024): scale[c] = gamma[c] rstd[c], shift[c] =
// beta[c] - mean[c] scale[c] (then y = far_affine_act_f32(x, scale, shift, residual, act)), mean_out / rstd_out for the backward,
// running_mean / running_var updated in place as nn.BatchNorm2d(momentum) does (either may be NULL; gamma / beta NULL = 1 / 0).
int far_bn_train_stats_f32(const float* x, long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out,
                           void* ws, long ws_bytes, hipStream_t stream) {
    far_clear_errors();
    if (!x || !scale || !shift || !mean_out || !rstd_out || !ws || M <= 0 || C <= 0 || (C & 3) || C > 1024 ||
        ws_bytes < far_bn_train_ws_bytes(M, C))
        return FAR_EINVAL;
    const int np = bn_parts(M);
    const long per = (M + np - 1) / np;
    hipLaunchKernelGGL(k_bn_partial<0>, dim3(np), dim3(BN_THREADS), 0, stream, (const float4*)x, (const float4*)nullptr,
                       (const float4*)nullptr, (const float*)nullptr, (const float*)nullptr, M, C / 4, per, 0, 0.f, (float*)ws);
    hipLaunchKernelGGL(k_bn_finalize_fwd, dim3((C + 63) / 64), dim3(BN_THREADS), 0, stream, (const float*)ws, np, M, per, C, gamma, beta, eps, momentum,
                       running_mean, running_var, scale, shift, mean_out, rstd_out);
    return far_check_launch();
}

// The whole training forward in one call: statistics as above, then y = act(x scale + shift (+ res)) (act 0 none, 1 ReLU, 2
// LeakyReLU(slope); res may be NULL).  vec: 4 C floats { scale, shift, mean, rstd } (kept by the caller for the backward).
int far_bn_act_train_fwd_f32(const float* x, const float* res, long M, int C, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, int act, float slope, float* y, float* vec,
                             void* ws, long ws_bytes, hipStream_t stream) {
    if (!y || !vec || act < 0 || act > 2) return FAR_EINVAL;
    const int rc = far_bn_train_stats_f32(x, M, C, gamma, beta, eps, momentum, running_mean, running_var, vec, vec + C, vec + 2 * C,
                                          vec + 3 * C, ws, ws_bytes, stream);
    if (rc != FAR_OK) return rc;
    const long nvec = M * (C / 4);
    long blocks = (nvec + 255) / 256;
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)blocks), dim3(256), 0, stream, (const float4*)x, (const float*)vec, (const float*)(vec + C),
                       (const float4*)res, nvec, C / 4, act, slope, (float4*)y);
    return far_check_launch();
}

// Backward of y = act(bn(x) (+ residual)) with batch statistics: dx, dgamma, dbeta, and dres = dy act'(y) when dres != NULL.
// x, dy, y: [M][C] channels_last fp32 (y only read when act != 0: 1 ReLU, 2 LeakyReLU(slope)); mean, rstd from the forward.
int far_bn_train_bwd_f32(const float* x, const float* dy, const float* y, const float* mean, const float* rstd, const float* gamma,
                         long M, int C, int act, float slope, float* dx, float* dgamma, float* dbeta, float* dres, void* ws,
                         long ws_bytes, hipStream_t stream) {
    far_clear_errors();
    if (!x || !dy || !mean || !rstd || !dx || !dgamma || !dbeta || !ws || M <= 0 || C <= 0 || (C & 3) || C > 1024 || act < 0 || act > 2 ||
        (act && !y) || ws_bytes < far_bn_train_ws_bytes(M, C))
        return FAR_EINVAL;
    const int np = bn_parts(M);
    const long per = (M + np - 1) / np;
    hipLaunchKernelGGL(k_bn_partial<1>, dim3(np), dim3(BN_THREADS), 0, stream, (const float4*)x, (const float4*)dy, (const float4*)y, mean,
                       rstd, M, C / 4, per, act, slope, (float*)ws);
    hipLaunchKernelGGL(k_bn_finalize_bwd, dim3((C + 63) / 64), dim3(BN_THREADS), 0, stream, (const float*)ws, np, C, dgamma, dbeta);
    const long nvec = M * (C / 4);
    long blocks = (nvec + 255) / 256;
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(k_bn_bwd_dx, dim3((unsigned)blocks), dim3(256), 0, stream, (const float4*)x, (const float4*)dy, (const float4*)y, mean,
                       rstd, gamma, (const float*)dgamma, (const float*)dbeta, nvec, C / 4, 1.0f / (float)M, act, slope, (float4*)dx,
                       (float4*)dres);
    return far_check_launch();
}

}  // extern "C"
