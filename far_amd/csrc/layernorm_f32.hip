// K6: LayerNorm over the channel dimension with optional fused residual add -- the two normalisations of every
// LoFTR encoder layer.
//
// Replaces mp3d_loftr/src/loftr/loftr_module/transformer.py:61   message = self.norm1(message)
//                                                             :65-67 message = self.norm2(message); return x + message
// (nn.LayerNorm(d_model), eps 1e-5, biased variance) and the LayerNorms of the head (:342, :346, :426).
// One wave per token row; the row lives in registers (C/64 floats per lane, 16-byte loads), mean and variance are
// two in-register passes (no E[x^2]-E[x]^2 cancellation), one read + one write of HBM per element.
#include "common.h"

namespace {

template <int VPL>   // float4 vectors per lane: C <= 256 * VPL, C % 4 == 0
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ res,
                                                   long rows, int C, float eps, float* __restrict__ y) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4* xr = reinterpret_cast<const float4*>(x + row * C);
    const int nvec = C >> 2;
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = (lane + 64 * i < nvec) ? xr[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += shfl_xor_f(s, d);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        if (lane + 64 * i < nvec) {
            float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) q += shfl_xor_f(q, d);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    const float4* r4 = res ? reinterpret_cast<const float4*>(res + row * C) : nullptr;
    float4* yr = reinterpret_cast<float4*>(y + row * C);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        if (lane + 64 * i >= nvec) continue;
        const float4 g = g4[lane + 64 * i], b = b4[lane + 64 * i];
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + b.x;
        o.y = (v[i].y - mean) * rstd * g.y + b.y;
        o.z = (v[i].z - mean) * rstd * g.z + b.z;
        o.w = (v[i].w - mean) * rstd * g.w + b.w;
        if (r4) { const float4 r = r4[lane + 64 * i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        yr[lane + 64 * i] = o;
    }
}

// C <= 256 (one float4 per lane): four rows per wave, their loads issued back to back -- a single 16-byte load per
// lane in flight does not cover the HBM latency.
__global__ __launch_bounds__(256) void k_layernorm_r4(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ res,
                                                      long rows, int C, float eps, float* __restrict__ y) {
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    const int lane = threadIdx.x & 63;
    const int nvec = C >> 2;
    const bool act = lane < nvec;
    float4 v[4], r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long row = row0 + k;
        const bool ok = act && row < rows;
        v[k] = ok ? reinterpret_cast<const float4*>(x + row * C)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        r[k] = (ok && res) ? reinterpret_cast<const float4*>(res + row * C)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 g = act ? reinterpret_cast<const float4*>(gamma)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = act ? reinterpret_cast<const float4*>(beta)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long row = row0 + k;
        float s = (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += shfl_xor_f(s, d);
        const float mean = s / (float)C;
        float q = 0.f;
        if (act) {
            const float a0 = v[k].x - mean, a1 = v[k].y - mean, a2 = v[k].z - mean, a3 = v[k].w - mean;
            q = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) q += shfl_xor_f(q, d);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        if (act && row < rows) {
            float4 o;
            o.x = (v[k].x - mean) * rstd * g.x + b.x;
            o.y = (v[k].y - mean) * rstd * g.y + b.y;
            o.z = (v[k].z - mean) * rstd * g.z + b.z;
            o.w = (v[k].w - mean) * rstd * g.w + b.w;
            if (res) { o.x += r[k].x; o.y += r[k].y; o.z += r[k].z; o.w += r[k].w; }
            reinterpret_cast<float4*>(y + row * C)[lane] = o;
        }
    }
}

// generic fallback for other channel counts (C % 4 == 0, C <= 4096): row cached in registers by strided float4
__global__ __launch_bounds__(256) void k_layernorm_any(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res,
                                                       long rows, int C, float eps, float* __restrict__ y) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += shfl_xor_f(s, d);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { float a = xr[c] - mean; q += a * a; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) q += shfl_xor_f(q, d);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    for (int c = lane; c < C; c += 64) {
        float o = (xr[c] - mean) * rstd * gamma[c] + beta[c];
        if (res) o += res[row * C + c];
        y[row * C + c] = o;
    }
}

}  // namespace

extern "C" {

// y[r][:] = LayerNorm(x[r][:]) * gamma + beta (+ res[r][:]);  x, res, y [rows][C] fp32 contiguous (y may alias x or res).
int far_layernorm_f32(const float* x, const float* gamma, const float* beta, const float* res, long rows, int C,
                      float eps, float* y, hipStream_t stream) {
    far_clear_errors();
    if (rows == 0) return FAR_OK;
    if (!x || !gamma || !beta || !y || rows < 0 || C <= 0) return FAR_EINVAL;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if ((C & 3) == 0 && C <= 256)
        hipLaunchKernelGGL(k_layernorm_r4, dim3((unsigned)((rows + 15) / 16)), block, 0, stream, x, gamma, beta, res, rows, C, eps, y);
    else if ((C & 3) == 0 && C <= 512) hipLaunchKernelGGL(k_layernorm<2>, grid, block, 0, stream, x, gamma, beta, res, rows, C, eps, y);
    else hipLaunchKernelGGL(k_layernorm_any, grid, block, 0, stream, x, gamma, beta, res, rows, C, eps, y);
    return far_check_launch();
}

}  // extern "C"
