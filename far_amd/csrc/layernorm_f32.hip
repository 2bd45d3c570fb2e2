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

// ---- backward (training): dx, and per-workgroup partial sums of dgamma / dbeta that k_ln_bwd_reduce adds in index order
// (deterministic).  One wave per row at a time, the row in registers (VPL float4 per lane: C <= 256 VPL); statistics recomputed
// exactly as the forward forms them (two in-register passes).
//   xh = (x - mean) rstd,  gg = dy gamma,  dx = rstd (gg - mean_c(gg) - xh mean_c(gg xh)),  dgamma = sum_rows dy xh,  dbeta = sum_rows dy
template <int VPL>
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                long rows, int C, float eps, float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[4][2][256 * VPL * 4 / 4];             // [wave][dgamma | dbeta][channel] for the cross-wave sum (C <= 1024)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2;
    const float invC = 1.0f / (float)C;
    float4 gm[VPL], dg[VPL], db[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bool a = lane + 64 * i < nvec;
        gm[i] = a ? reinterpret_cast<const float4*>(gamma)[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + row * C);
        const float4* gr = reinterpret_cast<const float4*>(dy + row * C);
        float4 v[VPL], g[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const bool a = lane + 64 * i < nvec;
            v[i] = a ? xr[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
            g[i] = a ? gr[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += shfl_xor_f(s, d);
        const float mean = s * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (lane + 64 * i < nvec) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (c * c + d * d);
            }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) q += shfl_xor_f(q, d);
        const float rstd = 1.0f / sqrtf(q * invC + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {                       // inactive lanes hold zeros in g: they add nothing
            v[i].x = (v[i].x - mean) * rstd; v[i].y = (v[i].y - mean) * rstd; v[i].z = (v[i].z - mean) * rstd; v[i].w = (v[i].w - mean) * rstd;
            dg[i].x += g[i].x * v[i].x; dg[i].y += g[i].y * v[i].y; dg[i].z += g[i].z * v[i].z; dg[i].w += g[i].w * v[i].w;
            db[i].x += g[i].x; db[i].y += g[i].y; db[i].z += g[i].z; db[i].w += g[i].w;
            g[i].x *= gm[i].x; g[i].y *= gm[i].y; g[i].z *= gm[i].z; g[i].w *= gm[i].w;
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { s1 += shfl_xor_f(s1, d); s2 += shfl_xor_f(s2, d); }
        const float m1 = s1 * invC, m2 = s2 * invC;
        float4* dr = reinterpret_cast<float4*>(dx + row * C);
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (lane + 64 * i < nvec)
                dr[lane + 64 * i] = make_float4(rstd * (g[i].x - m1 - v[i].x * m2), rstd * (g[i].y - m1 - v[i].y * m2),
                                                rstd * (g[i].z - m1 - v[i].z * m2), rstd * (g[i].w - m1 - v[i].w * m2));
    }
    // the four waves' sums, added in wave order
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        if (lane + 64 * i < nvec) {
            reinterpret_cast<float4*>(red[wave][0])[lane + 64 * i] = dg[i];
            reinterpret_cast<float4*>(red[wave][1])[lane + 64 * i] = db[i];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const int which = c >= C, cc = c - which * C;
        part[((size_t)blockIdx.x * 2 + which) * C + cc] = ((red[0][which][cc] + red[1][which][cc]) + red[2][which][cc]) + red[3][which][cc];
    }
}

// dgamma | dbeta [c] = sum over the workgroups' partials, in a fixed order: sixteen strided sums per channel (b = g, g + 16, ...)
// by sixteen threads, then added in g order.  64 channels per workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_ln_bwd_reduce(const float* __restrict__ part, int nblocks, int C, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta) {
    __shared__ float q[16][64];
    const int l = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    float s = 0.f;
    if (c < 2 * C) {
        const int which = c >= C, cc = c - which * C;
        for (int b = grp; b < nblocks; b += 16) s += part[((size_t)b * 2 + which) * C + cc];
    }
    q[grp][l] = s;
    __syncthreads();
    if (grp == 0 && c < 2 * C) {
        float t = q[0][l];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += q[g][l];
        if (c >= C) dbeta[c - C] = t; else dgamma[c] = t;
    }
}

inline int ln_bwd_blocks(long rows) {
    long b = (rows + 7) / 8;                                   // >= 2 rows per wave: enough waves to cover the load latency
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));             // <= 512 partials: the reduction reads nblocks x 2C floats
}

}  // namespace

extern "C" {

// Bytes of device scratch far_layernorm_bwd_f32 needs (0: shape not covered -- C % 4 != 0 or C > 1024).
long far_layernorm_bwd_ws_bytes(long rows, int C) {
    if (rows <= 0 || C <= 0 || (C & 3) || C > 1024) return 0;
    return (long)ln_bwd_blocks(rows) * 2 * C * (long)sizeof(float);
}

// Backward of y = LayerNorm(x; eps) * gamma + beta over the last dimension (autograd of nn.LayerNorm at
// transformer.py:61, 65-67): dx [rows][C], dgamma [C], dbeta [C] from x, gamma, dy; statistics recomputed as the forward forms
// them; dgamma / dbeta summed in a fixed order (deterministic).  C % 4 == 0, C <= 1024.
int far_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, long rows, int C, float eps, float* dx,
                          float* dgamma, float* dbeta, void* ws, long ws_bytes, hipStream_t stream) {
    far_clear_errors();
    const long need = far_layernorm_bwd_ws_bytes(rows, C);
    if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || !ws || need == 0 || ws_bytes < need) return FAR_EINVAL;
    const int nb = ln_bwd_blocks(rows);
    float* part = reinterpret_cast<float*>(ws);
    if (C <= 256) hipLaunchKernelGGL(k_ln_bwd<1>, dim3(nb), dim3(256), 0, stream, x, gamma, dy, rows, C, eps, dx, part);
    else if (C <= 512) hipLaunchKernelGGL(k_ln_bwd<2>, dim3(nb), dim3(256), 0, stream, x, gamma, dy, rows, C, eps, dx, part);
    else hipLaunchKernelGGL(k_ln_bwd<4>, dim3(nb), dim3(256), 0, stream, x, gamma, dy, rows, C, eps, dx, part);
    hipLaunchKernelGGL(k_ln_bwd_reduce, dim3((unsigned)((2 * C + 63) / 64)), dim3(1024), 0, stream, part, nb, C, dgamma, dbeta);
    return far_check_launch();
}


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
