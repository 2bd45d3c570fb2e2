// K15: the small dense layers of the regression head as row-independent exact-fp32 kernels.
//
// Replaces, for inference, the nn.Linear / torch.bmm calls of mp3d_loftr/src/loftr/loftr_module/transformer.py that were
// still vendor GEMMs:   :294-300  F = v~^T (P v~) (the 70 x N x 70 contraction after K2) and proj_fundamental
//                       :423-431  encoder (35840 -> 512 -> 512), pose_regressor_simple_moe (512 -> 512 -> 9)
//                       :448-458  moe_predictor (35862 -> 512 -> 512 -> 2, sigmoid)
// These are a few hundred MFLOP on B <= 32 rows: nothing for the matrix cores, but a vendor GEMM picks its kernel (tile,
// split-K factor) by the row count, so a pair's regressed pose depended on how many other pairs shared the batch (1e-7
// relative; the batch-32 test needed a 1e-4 bar).  Here every output is one fp32 fma chain over k in a FIXED order --
// the k range is cut into slices of a size that depends on K alone (slice_of), slice partials are summed in slice order -- so row r of a
// batch of 32 is bit-identical to the same pair run alone.  Memory-bound on the weights (35840 x 512 fp32 = 73 MB per big
// layer), which are packed once as [k / 4][n][4] so that a wave reads 1 KiB contiguous per load.
#include "common.h"

namespace {

constexpr int RB = 32;            // rows per launch block (zero-filled beyond B)
// k per slice: fixed per K (the summation order must not depend on the row count or on anything else): 256 for the two 35840-wide
// first layers, 64 for the 512- / 22-wide ones behind them -- those are 2 x 2 workgroups of one 256-k slice each otherwise, and a
// launch of theirs is one thread's chain of 8192 fmas (50-64 us, thirteen of them per step and per pair at batch 1)
__host__ __device__ constexpr int slice_of(int K) { return K > 2048 ? 256 : 64; }

// partial[s][b][n] = sum_{k in slice s} x[b][k] W[n][k].  RBT = row slots of the block (32, or 4 / 1 for small batches: a row's
// chain is its own, so fewer slots change no bit -- at batch 1 the 32-slot form spent 31 of 32 fmas on zero rows).
template <int RBT, int KSLT>
__global__ __launch_bounds__(256) void k_rows_partial(const float* __restrict__ x, long ldx, const float4* __restrict__ wp,
                                                      int B, int K, int N, float* __restrict__ partial) {
    constexpr int KT4 = KSLT >= 128 ? 32 : KSLT / 4;                // k per LDS tile / 4
    __shared__ float4 xs[RBT][KT4];
    const int s = blockIdx.x, n = blockIdx.y * 256 + threadIdx.x;
    const int k0 = s * KSLT;
    float acc[RBT];
#pragma unroll
    for (int b = 0; b < RBT; ++b) acc[b] = 0.f;
    const int K4 = (K + 3) >> 2;
    for (int sub = 0; sub < KSLT / (4 * KT4); ++sub) {
        const int kb = k0 + sub * 4 * KT4;
        __syncthreads();
        for (int i = threadIdx.x; i < RBT * KT4; i += 256) {
            const int b = i / KT4, q = i - b * KT4, k = kb + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < B) {
                const float* src = x + (long)b * ldx + k;
                if (k + 3 < K) { v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3]; }
                else { if (k < K) v.x = src[0]; if (k + 1 < K) v.y = src[1]; if (k + 2 < K) v.z = src[2]; }
            }
            xs[b][q] = v;
        }
        __syncthreads();
        if (n < N) {
#pragma unroll 4
            for (int q = 0; q < KT4; ++q) {
                const int k4 = (kb >> 2) + q;
                if (k4 >= K4) break;
                const float4 w = wp[(size_t)k4 * N + n];
#pragma unroll
                for (int b = 0; b < RBT; ++b) {
                    const float4 v = xs[b][q];                      // same address in every lane: an LDS broadcast
                    acc[b] = __builtin_fmaf(v.x, w.x, acc[b]);
                    acc[b] = __builtin_fmaf(v.y, w.y, acc[b]);
                    acc[b] = __builtin_fmaf(v.z, w.z, acc[b]);
                    acc[b] = __builtin_fmaf(v.w, w.w, acc[b]);
                }
            }
        }
    }
    if (n < N)
        for (int b = 0; b < B && b < RBT; ++b) partial[((size_t)s * B + b) * N + n] = acc[b];
}

// y[b][n] = act(bias[n] + add[b][n] + sum_s partial[s][b][n]), slices in order.  act: 0 none, 1 ReLU, 2 sigmoid, 3 GELU (erf)
__global__ void k_rows_reduce(const float* __restrict__ partial, const float* __restrict__ bias, const float* __restrict__ add,
                              long ld_add, int S, int B, int N, int act, float* __restrict__ y, long ldy) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * N) return;
    const int b = (int)(i / N), n = (int)(i - (long)b * N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += partial[((size_t)s * B + b) * N + n];
    if (bias) v += bias[n];
    if (add) v += add[(long)b * ld_add + n];
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 2) v = 1.0f / (1.0f + expf(-v));
    else if (act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    y[(long)b * ldy + n] = v;
}

// [N][K] (torch Linear weight) -> [ceil(K / 4)][N][4], zero padded
__global__ void k_rows_pack(const float* __restrict__ w, int N, int K, float4* __restrict__ wp) {
    const int K4 = (K + 3) >> 2;
    const long total = (long)K4 * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N), k = 4 * (int)(i / N);
        const float* src = w + (size_t)n * K + k;
        float4 v;
        v.x = k < K ? src[0] : 0.f; v.y = k + 1 < K ? src[1] : 0.f; v.z = k + 2 < K ? src[2] : 0.f; v.w = k + 3 < K ? src[3] : 0.f;
        wp[i] = v;
    }
}

// F[z][a][b] = sum_n vt[z][n][a] T[z][n][b]   (vt = [v | pos], T = P vt; a, b < 70), n in a FIXED order: the sequence is cut
// into NSEG segments of whole 64-token tiles (the same cut for every launch), a workgroup sums one segment with each thread
// holding a 4 x 5 block of F in registers (one 16-byte + one 16-byte + one 4-byte LDS read per 20 fmas), and the segment sums
// are added in segment order.
constexpr int DV = 70, NSEG = 4, VSTR = 72, TSTR = 72;
__global__ __launch_bounds__(256) void k_emm_contract(const float* __restrict__ v, long v_head_stride, long v_prob_stride, int heads,
                                                      const float* __restrict__ pos, const float* __restrict__ T, int N,
                                                      float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float vs[64][VSTR], ts[64][TSTR];
    const int z = blockIdx.x, seg = blockIdx.y, Z = gridDim.x;
    const int pz = z / heads, hh = z - pz * heads;
    const float* vz = v + (size_t)hh * v_head_stride + (size_t)pz * v_prob_stride;
    const float* Tz = T + (size_t)z * N * DV;
    const int ntile = (N + 63) / 64, per = (ntile + NSEG - 1) / NSEG;
    const int t0 = seg * per, t1 = min(ntile, t0 + per);
    const int ai = threadIdx.x / 14, bj = threadIdx.x - 14 * ai;       // a = 4 ai .. 4 ai + 3 (ai < 18), b = 5 bj .. 5 bj + 4
    const bool live = ai < 18;
    float acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = 0.f;
    for (int t = t0; t < t1; ++t) {
        const int n0 = t * 64;
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * VSTR; i += 256) {
            const int r = i / VSTR, c = i - r * VSTR, n = n0 + r;
            float vv = 0.f, tt = 0.f;
            if (n < N && c < DV) {
                vv = c < 64 ? vz[(size_t)n * 64 + c] : pos[(size_t)n * 6 + (c - 64)];
                tt = Tz[(size_t)n * DV + c];
            }
            vs[r][c] = vv; ts[r][c] = tt;
        }
        __syncthreads();
        if (live) {
#pragma unroll 4
            for (int r = 0; r < 64; ++r) {
                const float4 va = *reinterpret_cast<const float4*>(&vs[r][4 * ai]);
                float tb[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) tb[j] = ts[r][5 * bj + j];
                const float a4[4] = {va.x, va.y, va.z, va.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_fmaf(a4[i], tb[j], acc[i][j]);
            }
        }
    }
    if (live)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a = 4 * ai + i;
            if (a < DV)
#pragma unroll
                for (int j = 0; j < 5; ++j) part[(((size_t)seg * Z + z) * DV + a) * DV + 5 * bj + j] = acc[i][j];
        }
}

__global__ void k_emm_contract_sum(const float* __restrict__ part, long per_seg, float* __restrict__ F) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_seg) return;
    float v = part[i];
#pragma unroll
    for (int s = 1; s < NSEG; ++s) v += part[(size_t)s * per_seg + i];
    F[i] = v;
}

}  // namespace

extern "C" {

size_t far_rows_linear_packed_bytes(int N, int K) { return (N > 0 && K > 0) ? (size_t)((K + 3) / 4) * N * 16 : 0; }
size_t far_rows_linear_workspace_bytes(int B, int N, int K) {
    return (B > 0 && N > 0 && K > 0) ? (size_t)((K + slice_of(K) - 1) / slice_of(K)) * (B < RB ? B : RB) * N * 4 : 0;
}

// w [N][K] fp32 (torch nn.Linear layout) -> packed (far_rows_linear_packed_bytes bytes)
int far_rows_linear_pack_f32(const float* w, int N, int K, void* packed, hipStream_t stream) {
    far_clear_errors();
    if (!w || !packed || N <= 0 || K <= 0) return FAR_EINVAL;
    hipLaunchKernelGGL(k_rows_pack, dim3(1024), dim3(256), 0, stream, w, N, K, (float4*)packed);
    return far_check_launch();
}

// y[b][:] = act(x[b][:] W^T + bias + add[b][:]) for B rows (any B: processed 32 at a time); x rows ldx floats apart, y rows ldy,
// add (optional, e.g. the contribution of a second input block computed by another call) rows ld_add.  ws:
// far_rows_linear_workspace_bytes(B, N, K).  act: 0 none, 1 ReLU, 2 sigmoid, 3 GELU (erf form, nn.GELU's default).
// Row b's result does not depend on B or on the other rows (fixed summation order): see the header comment.
int far_rows_linear_f32(const float* x, long ldx, const void* packed, const float* bias, const float* add, long ld_add, int B,
                        int K, int N, int act, float* y, long ldy, void* ws, hipStream_t stream) {
    far_clear_errors();
    if (B == 0) return FAR_OK;
    if (!x || !packed || !y || !ws || B < 0 || K <= 0 || N <= 0 || act < 0 || act > 3 || ldx < K || ldy < N || (add && ld_add < N))
        return FAR_EINVAL;
    const int ksl = slice_of(K), S = (K + ksl - 1) / ksl;
    for (int b0 = 0; b0 < B; b0 += RB) {
        const int nb = B - b0 < RB ? B - b0 : RB;
        const dim3 grid(S, (N + 255) / 256);
        const float* xb = x + (long)b0 * ldx;
#define FAR_ROWS_LAUNCH(RBT, KSLT) hipLaunchKernelGGL((k_rows_partial<RBT, KSLT>), grid, dim3(256), 0, stream, xb, ldx, (const float4*)packed, nb, K, N, (float*)ws)
        if (ksl == 256) { if (nb == 1) FAR_ROWS_LAUNCH(1, 256); else if (nb <= 4) FAR_ROWS_LAUNCH(4, 256); else FAR_ROWS_LAUNCH(32, 256); }
        else { if (nb == 1) FAR_ROWS_LAUNCH(1, 64); else if (nb <= 4) FAR_ROWS_LAUNCH(4, 64); else FAR_ROWS_LAUNCH(32, 64); }
#undef FAR_ROWS_LAUNCH
        hipLaunchKernelGGL(k_rows_reduce, dim3((unsigned)(((long)nb * N + 255) / 256)), dim3(256), 0, stream, (const float*)ws, bias,
                           add ? add + (long)b0 * ld_add : nullptr, ld_add, S, nb, N, act, y + (long)b0 * ldy, ldy);
    }
    return far_check_launch();
}

// F [Z][70][70] = vt^T T per problem, vt = [v | pos] (v: problem z = p * heads + hh at v + hh * head_stride + p * prob_stride,
// [N][64] contiguous -- the planes far_emm_pv_f16s reads; pos [N][6]), T [Z][N][70] from far_emm_pv_*.  fp32 fma chains in a
// fixed order over n: a problem's F does not depend on how many problems share the launch (transformer.py:291-295).
// ws: far_emm_contract_workspace_bytes(Z) bytes.
size_t far_emm_contract_workspace_bytes(int Z) { return Z > 0 ? (size_t)NSEG * Z * DV * DV * 4 : 0; }
int far_emm_contract_f32(const float* v, int heads, long head_stride, long prob_stride, const float* pos, const float* T, int Z,
                         int N, float* F, void* ws, hipStream_t stream) {
    far_clear_errors();
    if (Z == 0) return FAR_OK;
    if (!v || !pos || !T || !F || !ws || Z < 0 || N <= 0 || heads < 1 || Z % heads) return FAR_EINVAL;
    hipLaunchKernelGGL(k_emm_contract, dim3(Z, NSEG), dim3(256), 0, stream, v, head_stride, prob_stride, heads, pos, T, N, (float*)ws);
    const long per_seg = (long)Z * DV * DV;
    hipLaunchKernelGGL(k_emm_contract_sum, dim3((unsigned)((per_seg + 255) / 256)), dim3(256), 0, stream, (const float*)ws, per_seg, F);
    return far_check_launch();
}

}  // extern "C"
