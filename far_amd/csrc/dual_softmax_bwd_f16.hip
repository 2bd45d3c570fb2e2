// K1 on the training path: the coarse loss of FAR's configuration reads conf_matrix ONLY at the ground-truth positions
// (sparse supervision + dual-softmax + focal loss: loftr_loss.py:86-112, `conf[pos_mask]`; positions = spv_b/i/j_ids,
// supervision.py:113-137).  So neither the 92 MB conf_matrix nor the 92 MB conf_matrix_gt of the reference exists here:
//
//   forward   far_coarse_pos_conf_f16s      statistics passes of the matcher (dual_softmax_f16s.hip) + the M values
//                                           p_k = conf[b_k, i_k, j_k]
//   backward  far_coarse_pos_conf_bwd_f16   given w_k = dL/dp_k * p_k:   with log p_k = 2 x_k - lse_row(i_k) - lse_col(j_k),
//                 dL/dx_ab = 2 W_ab - u_a R_ab - v_b C_ab,   u_a = sum_{k: i_k = a} w_k,  v_b = sum_{k: j_k = b} w_k,
//                 R = softmax over columns, C = softmax over rows, W the sparse matrix of the w_k;
//                 dF0 = kappa (dL/dx) F1,   dF1 = kappa (dL/dx)^T F0,   kappa = 1 / (C temperature).
//             The dense part G = u_a R_ab + v_b C_ab is never materialised: k1_bwd recomputes a 32 x 32 tile of x on the
//             f16 matrix core (transposed, so that a lane owns one row), forms G with two exp2 per entry from the forward's
//             statistics, and feeds it straight from the accumulator registers into the second MFMA (G F) as its A operand;
//             the other map's tile arrives channel-major ("transposed plane", columns pre-permuted into the order in which
//             the accumulator registers hold G, as K2 does for P v).  One kernel, launched twice with the roles of the two
//             maps swapped (dF0: rows = L side; dF1: rows = S side).  The sparse part 2 W is M scaled row additions.
//   Arithmetic: plain fp16 operands, fp32 accumulation, for both contractions (gradients: ~1e-3 relative, measured in
//   tests/test_train_kernels_gpu.py against float64 autograd); u, v are scaled by a power of two so that G <= 1 in fp16.
#include "dual_softmax_common.h"
#include <algorithm>

namespace far_k1b {

using namespace far_ds;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int C = 256;
constexpr int NS = C / 16;
constexpr int KT = 32;                   // columns per tile
constexpr int ROWB = C * 2;              // bytes per fp16 row of the row-major planes
constexpr int TILE_X = KT * ROWB;        // 16 KiB: tile of the row-major (swizzled) hi plane, for the score recompute
constexpr int TROW = 80;                 // bytes per channel row of a transposed tile: 32 positions x 2 B + 16 B pad
constexpr int TILE_T = C * TROW;         // 20 KiB
constexpr int STAGE = TILE_X + TILE_T;   // 36 KiB
constexpr float PRESCALE = 16.0f;
constexpr float HUGE_F = 1.0e30f;

// position of column c (0..31) inside a transposed tile row: MFMA step u = c >> 4 takes, from lane half h, the eight
// accumulator registers r = 8 u + e, which hold the columns 16 u + 4 h + (e & 3) + 8 (e >> 2)
__device__ __host__ inline int tpos(int c) {
    const int u = c >> 4, c16 = c & 15, h = (c16 >> 2) & 1, e = (c16 & 3) + 4 * (c16 >> 3);
    return (2 * u + h) * 8 + e;
}

// x [Z][N][256] fp32 -> transposed fp16 tiles [Z][Np / 32][256 ch][TROW], value * 2^4, rows >= N zero
__global__ void k1b_prep_t(const float* __restrict__ x, int Z, int N, int Np, unsigned char* __restrict__ out) {
    const long total = (long)Z * (Np / KT) * C * 4;                       // one thread = 8 positions of one channel row
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int q = (int)(t & 3);                                       // which 8 positions
        const int ch = (int)((t >> 2) & (C - 1));
        const long zt = t >> 10;                                          // z * ntile + tile
        const int ntile = Np / KT;
        const int jt = (int)(zt % ntile);
        const long z = zt / ntile;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // inverse of tpos for position p = 8 q + e:  u = q >> 1, h = q & 1, column = 16 u + 4 h + (e & 3) + 8 (e >> 2)
            const int c = 16 * (q >> 1) + 4 * (q & 1) + (e & 3) + 8 * (e >> 2);
            const int i = jt * KT + c;
            v[e] = i < N ? (_Float16)(x[((size_t)z * N + i) * C + ch] * PRESCALE) : (_Float16)0.f;
        }
        *reinterpret_cast<f16x8*>(out + (size_t)zt * TILE_T + ch * TROW + q * 16) = v;
    }
}

// p_k = conf at (b_k, i_k, j_k): one wave per position, float64 dot product of the fp32 features
__global__ __launch_bounds__(256) void k1b_pos_conf(const float* __restrict__ f0, const float* __restrict__ f1, int L, int S, int Sp,
                                                    double k2, const int64_t* __restrict__ pb, const int64_t* __restrict__ pi,
                                                    const int64_t* __restrict__ pj, int M, const float2* __restrict__ rowstat,
                                                    const float* __restrict__ cmax, const float* __restrict__ cinv,
                                                    float* __restrict__ p_out) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * (blockDim.x >> 6);
    for (int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < M; k += nw) {
        const size_t z = (size_t)pb[k], i = (size_t)pi[k], j = (size_t)pj[k];
        const float4 a = *reinterpret_cast<const float4*>(f0 + (z * L + i) * C + 4 * lane);
        const float4 b = *reinterpret_cast<const float4*>(f1 + (z * S + j) * C + 4 * lane);
        double d = (double)a.x * (double)b.x + (double)a.y * (double)b.y + (double)a.z * (double)b.z + (double)a.w * (double)b.w;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m, 64);
        if (lane == 0) {
            const float2 st = rowstat[z * L + i];
            const float x2 = (float)(2.0 * d * k2);
            p_out[k] = __builtin_amdgcn_exp2f((x2 - st.x) - cmax[z * Sp + j]) * (1.0f / st.y) * cinv[z * Sp + j];
        }
    }
}

// u[z L + i] += w_k, v[z S + j] += w_k   (fp32 atomics; M is a few thousand)
__global__ void k1b_scatter_uv(const int64_t* __restrict__ pb, const int64_t* __restrict__ pi, const int64_t* __restrict__ pj,
                               const float* __restrict__ w, int M, int L, int S, float* __restrict__ u, float* __restrict__ v,
                               unsigned* __restrict__ wmax_bits) {
    float mx = 0.f;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < M; k += gridDim.x * blockDim.x) {
        const float wk = w[k];
        atomicAdd(&u[(size_t)pb[k] * L + pi[k]], wk);
        atomicAdd(&v[(size_t)pb[k] * S + pj[k]], wk);
        mx = fmaxf(mx, fabsf(wk));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(wmax_bits, __float_as_uint(mx));      // non-negative floats order as uints
}

// scaled, padded weights and log-normalisers of one side:
//   wsc[z][Np] = u * 2^e (|u| sums of at most a few w: <= ~8 wmax -> e chosen so that 8 wmax 2^e <= 1), 0 past N
//   lnorm[z][Np] = max + log2(sum), +huge past N
__global__ void k1b_side(const float* __restrict__ u, const float2* __restrict__ stat, const float* __restrict__ dmax,
                         const float* __restrict__ dinv, int Z, int N, int Np, const unsigned* __restrict__ wmax_bits,
                         float* __restrict__ wsc, float* __restrict__ lnorm) {
    const float wmax = __uint_as_float(*wmax_bits);
    int e = 0;
    if (wmax > 0.f) { (void)frexpf(wmax, &e); e = -(e + 3); }
    const float sc = ldexpf(1.0f, e);
    const long total = (long)Z * Np;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = (int)(t % Np);
        const long z = t / Np;
        float wv = 0.f, ln = HUGE_F;
        if (i < N) {
            wv = u[z * N + i] * sc;
            if (stat) { const float2 st = stat[z * N + i]; ln = st.x + __builtin_amdgcn_logf(st.y); }
            else ln = dmax[z * Np + i] - __builtin_amdgcn_logf(dinv[z * Np + i]);
        }
        wsc[t] = wv;
        lnorm[t] = ln;
    }
}

__device__ __forceinline__ void dma_lin(unsigned char* lds, const unsigned char* g, int bytes, int tid, int wave) {
    for (int o = 0; o < bytes; o += 4096)
        __builtin_amdgcn_global_load_lds((gptr_t)(g + o + tid * 16), (lptr_t)(lds + o + wave * 1024), 16, 0, 0);
}

// out[z][row][256] = coef * sum_cols G[row][col] * B[col][:],   G = alpha_row 2^(x - rho_row) + beta_col 2^(x - gamma_col)
//   ah   row-side hi plane  [Z][Nrp][256] fp16 (swizzled LDS image of k1_prep)
//   bh   column-side hi plane, same layout; bt: column-side transposed tiles (k1b_prep_t)
// grid: Z * Nrp / 128 workgroups of 4 waves; wave = 32 rows x 256 channels of the output (128 accumulator registers)
__global__ __launch_bounds__(256, 2) void k1_bwd(const _Float16* __restrict__ ah, const _Float16* __restrict__ bh,
                                                 const unsigned char* __restrict__ bt, int Z, int Nr, int Nc, int Nrp, int Ncp,
                                                 float c1, const float* __restrict__ alpha, const float* __restrict__ rho,
                                                 const float* __restrict__ beta, const float* __restrict__ gamma,
                                                 const unsigned* __restrict__ wmax_bits, float kappa, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    int z, Ib;
    tile_coords(Nrp / 128, Z, z, Ib);
    const int irow = Ib * 128 + 32 * wave + l31;
    // row-side fragments for the score recompute (row irow, channels 16 s + 8 h ..): 64 VGPRs
    f16x8 af[NS];
    {
        const _Float16* p = ah + ((size_t)z * Nrp + irow) * C;
#pragma unroll
        for (int s = 0; s < NS; ++s) af[s] = *reinterpret_cast<const f16x8*>(p + 8 * ((2 * s + h) ^ (irow & 15)));
    }
    const float al = alpha[(size_t)z * Nrp + irow], rh = rho[(size_t)z * Nrp + irow];
    f32x16 acc[8];                                                // [channel block nt][rows]: D[m = row][n = channel]
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    float* const cw = reinterpret_cast<float*>(lds + 2 * STAGE);  // [2 stages][beta 32 | gamma 32]
    const int ntile = Ncp / KT;
    auto request = [&](int jt, int st) {
        unsigned char* base = lds + st * STAGE;
        dma_lin(base, reinterpret_cast<const unsigned char*>(bh + ((size_t)z * Ncp + (size_t)jt * KT) * C), TILE_X, tid, wave);
        dma_lin(base + TILE_X, bt + ((size_t)z * ntile + jt) * TILE_T, TILE_T, tid, wave);
        if (tid < KT) {
            cw[st * 64 + tid] = beta[(size_t)z * Ncp + jt * KT + tid];
            cw[st * 64 + 32 + tid] = gamma[(size_t)z * Ncp + jt * KT + tid];
        }
    };
    request(0, 0);
    for (int jt = 0; jt < ntile; ++jt) {
        const int st = jt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // tile jt landed; stage st ^ 1 is free
        if (jt + 1 < ntile) request(jt + 1, st ^ 1);
        const unsigned char* xs = lds + st * STAGE;
        // ---- scores, transposed: D[m = column of the tile][n = this lane's row]
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f16x8 bf = *reinterpret_cast<const f16x8*>(xs + l31 * ROWB + (((2 * s + h) ^ (l31 & 15)) * 16));
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf, af[s], sc, 0, 0, 0);
        }
        // ---- G for this lane's row and its 16 columns  c = (r & 3) + 8 (r >> 2) + 4 h
        f16x8 gp[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                             // streamed: four columns' (beta, gamma) at a time
            const float4 b4 = *reinterpret_cast<const float4*>(cw + st * 64 + 8 * q + 4 * h);
            const float4 g4 = *reinterpret_cast<const float4*>(cw + st * 64 + 32 + 8 * q + 4 * h);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q + e;
                const float x = sc[r] * c1;
                const float g = al * __builtin_amdgcn_exp2f(x - rh) + bb[e] * __builtin_amdgcn_exp2f(x - gg[e]);
                gp[r >> 3][r & 7] = (_Float16)g;
            }
        }
        // ---- out[row][channel] += G[row][col] * B[col][channel]: A = G (registers), B = transposed tile (LDS)
        const unsigned char* ts = xs + TILE_X;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const f16x8 tf = *reinterpret_cast<const f16x8*>(ts + (32 * nt + l31) * TROW + (2 * u + h) * 16);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gp[u], tf, acc[nt], 0, 0, 0);
            }
    }
    // ---- epilogue: undo the scalings (operands x 2^4, weights x 2^e) and apply -kappa
    const float wmax = __uint_as_float(*wmax_bits);
    int e = 0;
    if (wmax > 0.f) { (void)frexpf(wmax, &e); e = -(e + 3); }
    const float coef = -kappa * ldexpf(1.0f, -e) / PRESCALE;
    const int row0 = Ib * 128 + 32 * wave;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = row0 + mfma32_row(r, h);
            if (i < Nr) out[((size_t)z * Nr + i) * C + 32 * nt + l31] = acc[nt][r] * coef;
        }
}

// the sparse part: dF0[b, i_k] += 2 kappa w_k F1[b, j_k],  dF1[b, j_k] += 2 kappa w_k F0[b, i_k]  (one wave per position)
__global__ __launch_bounds__(256) void k1b_sparse(const float* __restrict__ f0, const float* __restrict__ f1, int L, int S,
                                                  const int64_t* __restrict__ pb, const int64_t* __restrict__ pi,
                                                  const int64_t* __restrict__ pj, const float* __restrict__ w, int M, float kappa2,
                                                  float* __restrict__ df0, float* __restrict__ df1) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * (blockDim.x >> 6);
    for (int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < M; k += nw) {
        const size_t r0 = ((size_t)pb[k] * L + pi[k]) * C, r1 = ((size_t)pb[k] * S + pj[k]) * C;
        const float s = kappa2 * w[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            atomicAdd(&df0[r0 + 64 * c + lane], s * f1[r1 + 64 * c + lane]);
            atomicAdd(&df1[r1 + 64 * c + lane], s * f0[r0 + 64 * c + lane]);
        }
    }
}

struct WsB {
    unsigned char *at, *bt;          // transposed tiles of f0 / f1
    float *u, *v;                    // [Z L], [Z S]
    float *alpha, *rho, *beta, *gamma;   // [Z Lp] x2, [Z Sp] x2
    unsigned* wmax;
    size_t bytes;
};
inline WsB carve_b(unsigned char* p, size_t o, int Z, int L, int S) {
    WsB w;
    const int Lp = (L + 127) / 128 * 128, Sp = (S + 127) / 128 * 128;
    auto take = [&](size_t n) { unsigned char* r = p ? p + o : nullptr; o += align256(n); return r; };
    w.at = take((size_t)Z * (Lp / KT) * TILE_T);
    w.bt = take((size_t)Z * (Sp / KT) * TILE_T);
    w.u = (float*)take((size_t)Z * L * 4); w.v = (float*)take((size_t)Z * S * 4);
    w.alpha = (float*)take((size_t)Z * Lp * 4); w.rho = (float*)take((size_t)Z * Lp * 4);
    w.beta = (float*)take((size_t)Z * Sp * 4); w.gamma = (float*)take((size_t)Z * Sp * 4);
    w.wmax = (unsigned*)take(256);
    w.bytes = o;
    return w;
}

}  // namespace far_k1b

// ---- glue with dual_softmax_f16s.hip (same translation-unit-independent layout of the forward workspace) ----
size_t far_k1_fwd_ws_bytes(int Z, int L, int S);
int far_k1_stats_launch(const float* f0, const float* f1, int Z, int L, int S, float temperature, void* ws, int* overflow,
                        hipStream_t stream);
void far_k1_fwd_views(void* ws, int Z, int L, int S, const _Float16** ah, const _Float16** bh, const float2** rowstat,
                      const float** cmax, const float** cinv, float* c1);

extern "C" {

size_t far_coarse_train_workspace_bytes(int Z, int L, int S, int Cc) {
    if (Z <= 0 || L <= 0 || S <= 0 || Cc != far_k1b::C) return 0;
    const size_t fwd = far_k1_fwd_ws_bytes(Z, L, S);
    return far_k1b::carve_b(nullptr, fwd, Z, L, S).bytes;
}

// Forward of the sparse coarse supervision: p_out[k] = conf_matrix[pb[k], pi[k], pj[k]] (coarse_matching.py:108-118 at the
// positions loftr_loss.py:86-91 reads), leaving operand planes + statistics in `ws` for the backward call.
int far_coarse_pos_conf_f16s(const float* f0, const float* f1, int Z, int L, int S, int Cc, float temperature,
                             const int64_t* pb, const int64_t* pi, const int64_t* pj, int M, float* p_out, void* ws,
                             int* overflow, hipStream_t stream) {
    using namespace far_k1b;
    far_clear_errors();
    if (!f0 || !f1 || !ws || Z <= 0 || L <= 0 || S <= 0 || Cc != C || M < 0 || (M > 0 && (!pb || !pi || !pj || !p_out)))
        return FAR_EINVAL;
    int rc = far_k1_stats_launch(f0, f1, Z, L, S, temperature, ws, overflow, stream);
    if (rc != FAR_OK) return rc;
    if (M > 0) {
        const _Float16 *ah, *bh;
        const float2* rowstat;
        const float *cmax, *cinv;
        float c1;
        far_k1_fwd_views(ws, Z, L, S, &ah, &bh, &rowstat, &cmax, &cinv, &c1);
        const int Sp = (S + 127) / 128 * 128;
        const double k2 = 1.4426950408889634 / ((double)C * (double)temperature);
        hipLaunchKernelGGL(k1b_pos_conf, dim3(std::min((M + 3) / 4, 2048)), dim3(256), 0, stream, f0, f1, L, S, Sp, k2, pb, pi, pj, M,
                           rowstat, cmax, cinv, p_out);
    }
    return far_check_launch();
}

// Backward: w[k] = dL/dp_k * p_k (the caller multiplies; for the focal loss this is bounded even where p -> 0).
// df0 (Z, L, C) and df1 (Z, S, C) are OVERWRITTEN with dL/dF0, dL/dF1.  `ws` must be the buffer the forward call left.
int far_coarse_pos_conf_bwd_f16(const float* f0, const float* f1, int Z, int L, int S, int Cc, float temperature,
                                const int64_t* pb, const int64_t* pi, const int64_t* pj, int M, const float* w, float* df0,
                                float* df1, void* ws, hipStream_t stream) {
    using namespace far_k1b;
    far_clear_errors();
    if (!f0 || !f1 || !ws || !df0 || !df1 || Z <= 0 || L <= 0 || S <= 0 || Cc != C || M < 0 || (M > 0 && (!pb || !pi || !pj || !w)))
        return FAR_EINVAL;
    const WsB b = carve_b((unsigned char*)ws, far_k1_fwd_ws_bytes(Z, L, S), Z, L, S);
    const int Lp = (L + 127) / 128 * 128, Sp = (S + 127) / 128 * 128;
    const _Float16 *ah, *bh;
    const float2* rowstat;
    const float *cmax, *cinv;
    float c1;
    far_k1_fwd_views(ws, Z, L, S, &ah, &bh, &rowstat, &cmax, &cinv, &c1);
    c1 = (float)(1.4426950408889634 / ((double)C * (double)temperature * PRESCALE * PRESCALE));   // log2-domain score per unit dot
    auto gridp = [](long n) { long g = (n + 255) / 256; return (unsigned)(g < 16384 ? (g > 0 ? g : 1) : 16384); };
    hipMemsetAsync(b.u, 0, (size_t)Z * L * 4, stream);
    hipMemsetAsync(b.v, 0, (size_t)Z * S * 4, stream);
    hipMemsetAsync(b.wmax, 0, 4, stream);
    if (M > 0) hipLaunchKernelGGL(k1b_scatter_uv, dim3(gridp(M)), dim3(256), 0, stream, pb, pi, pj, w, M, L, S, b.u, b.v, b.wmax);
    hipLaunchKernelGGL(k1b_side, dim3(gridp((long)Z * Lp)), dim3(256), 0, stream, b.u, rowstat, (const float*)nullptr,
                       (const float*)nullptr, Z, L, Lp, b.wmax, b.alpha, b.rho);
    hipLaunchKernelGGL(k1b_side, dim3(gridp((long)Z * Sp)), dim3(256), 0, stream, b.v, (const float2*)nullptr, cmax, cinv, Z, S, Sp,
                       b.wmax, b.beta, b.gamma);
    hipLaunchKernelGGL(k1b_prep_t, dim3(gridp((long)Z * (Lp / KT) * C * 4)), dim3(256), 0, stream, f0, Z, L, Lp, b.at);
    hipLaunchKernelGGL(k1b_prep_t, dim3(gridp((long)Z * (Sp / KT) * C * 4)), dim3(256), 0, stream, f1, Z, S, Sp, b.bt);
    const size_t smem = 2 * STAGE + 2 * 64 * sizeof(float);
    FAR_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)k1_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const float kappa = (float)(1.0 / ((double)C * (double)temperature));
    // dF0: rows = L side (u, row normalisers), columns = S side (v, column normalisers)
    hipLaunchKernelGGL(k1_bwd, dim3((Lp / 128) * Z), dim3(256), smem, stream, ah, bh, b.bt, Z, L, S, Lp, Sp, c1, b.alpha, b.rho, b.beta,
                       b.gamma, b.wmax, kappa, df0);
    // dF1: roles swapped
    hipLaunchKernelGGL(k1_bwd, dim3((Sp / 128) * Z), dim3(256), smem, stream, bh, ah, b.at, Z, S, L, Sp, Lp, c1, b.beta, b.gamma, b.alpha,
                       b.rho, b.wmax, kappa, df1);
    if (M > 0)
        hipLaunchKernelGGL(k1b_sparse, dim3(std::min((M + 3) / 4, 2048)), dim3(256), 0, stream, f0, f1, L, S, pb, pi, pj, w, M,
                           2.0f * kappa, df0, df1);
    return far_check_launch();
}

}  // extern "C"
