// K2 backward (training path, BASELINE configs[2]): gradients of the EMM head's bilinear dual-softmax attention
//     F = v~^T P v~,   P = softmax(s, keys) * softmax(s, queries),   s = (q k^T) * scale,   v~ = [v | pos]
// (mp3d_loftr/src/loftr/loftr_module/transformer.py:275-292) w.r.t. q and k, without the (B, 4, 4800, 4800) tensors the
// reference's autograd keeps (8 x 92 MB per pair forward, as many again saved for backward).
//
// With dF the upstream gradient, A = v~ dF, B = v~ dF^T (N x 70), T = P v~ (the forward's output), T' = P^T v~:
//     dP_ab = A_a . v~_b                      u_a = A_a . T_a      v_b = B_b . T'_b          (row / column sums of dP * P)
//     ds_ab = 2 P_ab dP_ab - R_ab u_a - C_ab v_b          (R: softmax over keys, C: softmax over queries, P = R C)
//     dq = scale ds k,   dk = scale ds^T q,   dv~ = T dF^T + T' dF
// T' is the forward kernel with q and k exchanged; A, B, u, v, dv~ are (N x 70)-sized torch expressions (far_amd/ops/head.py).
// What needs a kernel is ds contracted with k (and ds^T with q): k2_bwd recomputes a 32 x 32 tile of s AND of dP on the f16
// matrix core (transposed: a lane owns one row), forms ds from the forward's statistics with two exp2 per entry, and feeds
// it from the accumulator registers into the third MFMA as its A operand (other side's tile channel-major, columns
// permuted into register order -- the scheme of K1's backward, dual_softmax_bwd_f16.hip).  One kernel, two launches:
//     dq: rows = queries (x-operand q, dP-operand A, weight u, row normaliser), columns = keys (k, v~, v, column normaliser)
//     dk: rows = keys    (k, v~, v, column normaliser),                        columns = queries (q, A, u, row normaliser)
// Arithmetic: plain fp16 operands, fp32 accumulation (gradient-grade, ~1e-3; measured in tests/test_train_kernels_gpu.py
// against float64 autograd).  A, B, u, v arrive pre-scaled by the caller so that max |A|, |B| ~ 1.
#include "common.h"
#include <algorithm>

namespace far_k2b {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int D = 64;                 // head dim (x-operands q, k)
constexpr int DV = 70;                // v~ / A width
constexpr int DVP = 80;               // padded to 5 MFMA k-steps
constexpr int KT = 32;                // columns per tile
constexpr int XROW = 144;             // bytes per row of an x plane: 64 fp16 + 16 B pad (conflict-free ds_read_b128)
constexpr int YROW = 176;             // bytes per row of a dP-operand plane: 80 fp16 + 16 B pad
constexpr int TROW = 80;              // bytes per channel row of a transposed x tile: 32 positions + 16 B pad
constexpr int XT_TILE = D * TROW;     // 5 KiB
constexpr int X_SLOT = 5 * 1024, Y_SLOT = 6 * 1024, T_SLOT = 5 * 1024;      // LDS slots (DMA granule 1 KiB per wave)
constexpr int STAGE = X_SLOT + Y_SLOT + T_SLOT + 256;                      // + 32 (weight, normaliser) pairs
constexpr float PRE = 16.0f;
constexpr float HUGE_F = 1.0e30f;

// x [Z][N][W] fp32 (row stride `ld` floats) -> [Z][Np][rowbytes] fp16 rows (value * 2^4, W..Wp-1 and rows >= N zero)
__global__ void k2b_prep_rows(const float* __restrict__ x, int Z, int N, int Np, int W, int ld, int Wp, int rowbytes,
                              unsigned char* __restrict__ out) {
    const int nslot = Wp / 8;
    const long total = (long)Z * Np * nslot;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int slot = (int)(t % nslot);
        const long row = t / nslot;
        const int i = (int)(row % Np);
        const long z = row / Np;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * slot + e;
            v[e] = (i < N && c < W) ? (_Float16)(x[((size_t)z * N + i) * ld + c] * PRE) : (_Float16)0.f;
        }
        *reinterpret_cast<f16x8*>(out + (size_t)row * rowbytes + slot * 16) = v;
    }
}

// x [Z][N][64] fp32 -> transposed tiles [Z][Np / 32][64 ch][TROW]; position (2 u + h) 8 + e holds column
// 16 u + 4 h + (e & 3) + 8 (e >> 2) of the tile (the order in which the accumulator registers hold ds)
__global__ void k2b_prep_t(const float* __restrict__ x, int Z, int N, int Np, unsigned char* __restrict__ out) {
    const long total = (long)Z * (Np / KT) * D * 4;
    const int ntile = Np / KT;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int q = (int)(t & 3);
        const int ch = (int)((t >> 2) & (D - 1));
        const long zt = t >> 8;
        const int jt = (int)(zt % ntile);
        const long z = zt / ntile;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 16 * (q >> 1) + 4 * (q & 1) + (e & 3) + 8 * (e >> 2);
            const int i = jt * KT + c;
            v[e] = i < N ? (_Float16)(x[((size_t)z * N + i) * D + ch] * PRE) : (_Float16)0.f;
        }
        *reinterpret_cast<f16x8*>(out + (size_t)zt * XT_TILE + ch * TROW + q * 16) = v;
    }
}

// (weight, log-normaliser) per token, padded: wl[z][Np] = (w, max + log2 sum); past N: (0, +huge)
__global__ void k2b_side(const float* __restrict__ w, const float2* __restrict__ stat, int Z, int N, int Np, float2* __restrict__ wl) {
    const long total = (long)Z * Np;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = (int)(t % Np);
        const long z = t / Np;
        float2 o = make_float2(0.f, HUGE_F);
        if (i < N) {
            const float2 st = stat[z * N + i];
            o = make_float2(w[z * N + i], st.x + __builtin_amdgcn_logf(st.y));
        }
        wl[t] = o;
    }
}

// linear LDS-DMA of `bytes` (any multiple of 16; the last 1 KiB granule may run past the tile: the LDS slots and the
// global planes carry slack for it)
__device__ __forceinline__ void dma_lin(unsigned char* lds, const unsigned char* g, int bytes, int tid, int wave) {
    for (int o = 0; o < bytes; o += 4096)
        if (o + wave * 1024 < bytes)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + o + tid * 16), (lptr_t)(lds + o + wave * 1024), 16, 0, 0);
}

// out[z][row][64] = coef * sum_col ds[row][col] * Xcol[col][:]
__global__ __launch_bounds__(256, 2) void k2_bwd(const unsigned char* __restrict__ xr, const unsigned char* __restrict__ yr,
                                                 const float2* __restrict__ wlr, const unsigned char* __restrict__ xc,
                                                 const unsigned char* __restrict__ yc, const unsigned char* __restrict__ xct,
                                                 const float2* __restrict__ wlc, int Z, int N, int Np, float c1, float dps,
                                                 float coef, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    int z, Ib;
    tile_coords(Np / 128, Z, z, Ib);
    const int irow = Ib * 128 + 32 * wave + l31;
    f16x8 xf[4], yf[5];
    {
        const unsigned char* px = xr + ((size_t)z * Np + irow) * XROW;
        const unsigned char* py = yr + ((size_t)z * Np + irow) * YROW;
#pragma unroll
        for (int s = 0; s < 4; ++s) xf[s] = *reinterpret_cast<const f16x8*>(px + (2 * s + h) * 16);
#pragma unroll
        for (int s = 0; s < 5; ++s) yf[s] = *reinterpret_cast<const f16x8*>(py + (2 * s + h) * 16);
    }
    const float2 wl = wlr[(size_t)z * Np + irow];
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int ntile = Np / KT;
    auto request = [&](int jt, int st) {
        unsigned char* base = lds + st * STAGE;
        dma_lin(base, xc + ((size_t)z * Np + (size_t)jt * KT) * XROW, KT * XROW, tid, wave);
        dma_lin(base + X_SLOT, yc + ((size_t)z * Np + (size_t)jt * KT) * YROW, KT * YROW, tid, wave);
        dma_lin(base + X_SLOT + Y_SLOT, xct + ((size_t)z * ntile + jt) * XT_TILE, XT_TILE, tid, wave);
        if (tid < KT) reinterpret_cast<float2*>(base + X_SLOT + Y_SLOT + T_SLOT)[tid] = wlc[(size_t)z * Np + jt * KT + tid];
    };
    request(0, 0);
    for (int jt = 0; jt < ntile; ++jt) {
        const int st = jt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (jt + 1 < ntile) request(jt + 1, st ^ 1);
        const unsigned char* xs = lds + st * STAGE;
        const unsigned char* ys = xs + X_SLOT;
        const unsigned char* ts = ys + Y_SLOT;
        const float2* cw = reinterpret_cast<const float2*>(ts + T_SLOT);
        // scores and dP, transposed: D[m = column of the tile][n = this lane's row]
        f32x16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s)
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(xs + l31 * XROW + (2 * s + h) * 16), xf[s], sc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 5; ++s)
            dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(ys + l31 * YROW + (2 * s + h) * 16), yf[s], dp, 0, 0, 0);
        // ds for this lane's row and its 16 columns c = (r & 3) + 8 (r >> 2) + 4 h
        f16x8 gp[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float2 c = cw[mfma32_row(r, h)];
            const float x = sc[r] * c1;
            const float er = __builtin_amdgcn_exp2f(x - wl.y), ec = __builtin_amdgcn_exp2f(x - c.y);
            const float g = 2.0f * (er * ec) * (dp[r] * dps) - er * wl.x - ec * c.x;
            gp[r >> 3][r & 7] = (_Float16)g;
        }
        // out[row][channel] += ds[row][col] * Xcol[col][channel]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    gp[u], *reinterpret_cast<const f16x8*>(ts + (32 * nt + l31) * TROW + (2 * u + h) * 16), acc[nt], 0, 0, 0);
    }
    const int row0 = Ib * 128 + 32 * wave;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = row0 + mfma32_row(r, h);
            if (i < N) out[((size_t)z * N + i) * D + 32 * nt + l31] = acc[nt][r] * coef;
        }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct Ws {
    unsigned char *qx, *kx, *ay, *vy, *qt, *kt;
    float2 *wlq, *wlk;
    size_t bytes;
};
inline Ws carve(void* base, int Z, int N) {
    const int Np = (N + 127) / 128 * 128;
    unsigned char* p = (unsigned char*)base;
    size_t o = 0;
    auto take = [&](size_t n) { unsigned char* r = p ? p + o : nullptr; o += align256(n + 1024); return r; };   // + DMA slack
    Ws w;
    w.qx = take((size_t)Z * Np * XROW); w.kx = take((size_t)Z * Np * XROW);
    w.ay = take((size_t)Z * Np * YROW); w.vy = take((size_t)Z * Np * YROW);
    w.qt = take((size_t)Z * (Np / KT) * XT_TILE); w.kt = take((size_t)Z * (Np / KT) * XT_TILE);
    w.wlq = (float2*)take((size_t)Z * Np * 8); w.wlk = (float2*)take((size_t)Z * Np * 8);
    w.bytes = o;
    return w;
}

}  // namespace far_k2b

extern "C" {

size_t far_emm_bwd_workspace_bytes(int Z, int N) {
    if (Z <= 0 || N <= 0) return 0;
    return far_k2b::carve(nullptr, Z, N).bytes;
}

// dq, dk [Z][N][64] (overwritten) of the EMM bilinear attention.  q, k [Z][N][64]; vt = [v | pos] and A = vt dF
// [Z][N][70]; u [Z][N] = rowdot(A, T), vw [Z][N] = rowdot(vt dF^T, P^T vt); rowstat / colstat [Z][N] (max, sum) in the
// log2 domain, as far_emm_pv_f16s computed them (far_emm_pv_f16s_copy_stats).  A, u, vw may carry a common power-of-two
// scale (the caller divides dq, dk by it).
int far_emm_bwd_f16(const float* q, const float* k, const float* vt, const float* A, const float* u, const float* vw,
                    const float* rowstat, const float* colstat, int Z, int N, float scale, float* dq, float* dk, void* ws,
                    hipStream_t stream) {
    using namespace far_k2b;
    far_clear_errors();
    if (!q || !k || !vt || !A || !u || !vw || !rowstat || !colstat || !dq || !dk || !ws || Z <= 0 || N <= 0) return FAR_EINVAL;
    const Ws w = carve(ws, Z, N);
    const int Np = (N + 127) / 128 * 128;
    auto gridp = [](long n) { long g = (n + 255) / 256; return (unsigned)(g < 65536 ? (g > 0 ? g : 1) : 65536); };
    hipLaunchKernelGGL(k2b_prep_rows, dim3(gridp((long)Z * Np * 8)), dim3(256), 0, stream, q, Z, N, Np, D, D, D, XROW, w.qx);
    hipLaunchKernelGGL(k2b_prep_rows, dim3(gridp((long)Z * Np * 8)), dim3(256), 0, stream, k, Z, N, Np, D, D, D, XROW, w.kx);
    hipLaunchKernelGGL(k2b_prep_rows, dim3(gridp((long)Z * Np * 10)), dim3(256), 0, stream, A, Z, N, Np, DV, DV, DVP, YROW, w.ay);
    hipLaunchKernelGGL(k2b_prep_rows, dim3(gridp((long)Z * Np * 10)), dim3(256), 0, stream, vt, Z, N, Np, DV, DV, DVP, YROW, w.vy);
    hipLaunchKernelGGL(k2b_prep_t, dim3(gridp((long)Z * (Np / KT) * D * 4)), dim3(256), 0, stream, q, Z, N, Np, w.qt);
    hipLaunchKernelGGL(k2b_prep_t, dim3(gridp((long)Z * (Np / KT) * D * 4)), dim3(256), 0, stream, k, Z, N, Np, w.kt);
    hipLaunchKernelGGL(k2b_side, dim3(gridp((long)Z * Np)), dim3(256), 0, stream, u, (const float2*)rowstat, Z, N, Np, w.wlq);
    hipLaunchKernelGGL(k2b_side, dim3(gridp((long)Z * Np)), dim3(256), 0, stream, vw, (const float2*)colstat, Z, N, Np, w.wlk);
    const size_t smem = 2 * STAGE;
    FAR_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)k2_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const float c1 = scale * 1.44269504088896341f / (PRE * PRE);       // log2-domain score per unit of the pre-scaled dot
    const float dps = 1.0f / (PRE * PRE);                               // dP per unit of the pre-scaled dot
    const float coef = scale / PRE;                                     // ds Xcol / 2^4, times the score scale
    const dim3 grid((unsigned)(Np / 128) * Z);
    hipLaunchKernelGGL(k2_bwd, grid, dim3(256), smem, stream, w.qx, w.ay, w.wlq, w.kx, w.vy, w.kt, w.wlk, Z, N, Np, c1, dps, coef, dq);
    hipLaunchKernelGGL(k2_bwd, grid, dim3(256), smem, stream, w.kx, w.vy, w.wlk, w.qx, w.ay, w.qt, w.wlq, Z, N, Np, c1, dps, coef, dk);
    return far_check_launch();
}

}  // extern "C"
