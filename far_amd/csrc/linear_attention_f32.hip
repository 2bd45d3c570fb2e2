// K5: the linear-attention core of the LoFTR encoder layers.
//
// Replaces mp3d_loftr/src/loftr/loftr_module/linear_attention.py:31-50 (LinearAttention.forward):
//   Q = elu(q)+1, K = elu(k)+1; values = v / S
//   KV = einsum(nshd,nshv->nhdv, K, values);  Z = 1 / (einsum(nlhd,nhd->nlh, Q, K.sum(1)) + eps)
//   out = einsum(nlhd,nhdv,nlh->nlhv, Q, KV, Z) * S
// Inputs are the raw projections q [N][L][H*D], k, v [N][S][H*D]; the feature map (elu+1), the 1/S and *S
// scalings and the normaliser are fused.  Two streaming kernels (KV / K.sum per token chunk, then apply) plus a
// tiny fixed-order reduction over chunks (deterministic, no atomics).
#include "common.h"

namespace {

__device__ __forceinline__ float elu1(float x) { return (x > 0.f ? x : expm1f(x)) + 1.f; }  // F.elu(x) + 1

// ---------------------------------------------------------------------------------------------------------
// Both kernels put the per-head 32x32 (or 16x16) contractions on the exact-f32 matrix core so that what is left
// is the HBM stream of q / k / v / out:
//   D = 32: v_mfma_f32_32x32x2_f32   (lane = (col = lane & 31, kk = lane >> 5), 2 tokens or 2 channels per MFMA)
//   D = 16: v_mfma_f32_16x16x4_f32   (lane = (col = lane & 15, kk = lane >> 4), 4 per MFMA)
// ---------------------------------------------------------------------------------------------------------
template <int D> struct Mf;
template <> struct Mf<32> {
    typedef f32x16 acc_t;
    static constexpr int NACC = 16, KPER = 2, SHIFT = 5;
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int r, int kk) { return (r & 3) + 8 * (r >> 2) + 4 * kk; }
};
template <> struct Mf<16> {
    typedef f32x4 acc_t;
    static constexpr int NACC = 4, KPER = 4, SHIFT = 4;
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int r, int kk) { return 4 * kk + r; }
};

// KV[h][d][v] = sum_s K'[s][h,d] * (V[s][h,v] / S),  ksum[h][d] = sum_s K'[s][h,d]  over one token chunk.
// One wave per (n, chunk, head): D[m = d][n = v] += A[d][k = token] B[k = token][v]; both operands are read
// straight from global memory with the head's D channels across the lanes (2 x 128 B or 4 x 64 B per load).
// part: [N][nchunk][H*D][D+1] (last column = ksum).
template <int D>
__global__ __launch_bounds__(256) void k_la_kv_partial(const float* __restrict__ k, const float* __restrict__ v,
                                                       const uint8_t* __restrict__ kv_mask, int N, int S, int H,
                                                       int tok_per_chunk, int nchunk, float* __restrict__ part) {
    typedef Mf<D> M;
    const int lane = threadIdx.x & 63, col = lane & (D - 1), kk = lane >> M::SHIFT;
    const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long units = (long)N * nchunk * H;
    if (unit >= units) return;
    const int h = (int)(unit % H);
    const int ch = (int)((unit / H) % nchunk);
    const int n = (int)(unit / ((long)H * nchunk));
    const int HD = H * D;
    const int s0 = ch * tok_per_chunk, s1 = min(S, s0 + tok_per_chunk);
    const float fS = (float)S;
    typename M::acc_t acc;
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) acc[r] = 0.f;
    float ks = 0.f;
    const float* kp = k + ((size_t)n * S) * HD + h * D + col;
    const float* vp = v + ((size_t)n * S) * HD + h * D + col;
    constexpr int U = 8;   // token groups in flight
    for (int sb = s0; sb < s1; sb += U * M::KPER) {
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = sb + u * M::KPER + kk;
            a[u] = 0.f; b[u] = 0.f;
            if (s < s1) { a[u] = kp[(size_t)s * HD]; b[u] = vp[(size_t)s * HD]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = sb + u * M::KPER + kk;
            float m = 1.f;
            if (kv_mask && s < s1) m = kv_mask[(size_t)n * S + s] ? 1.f : 0.f;
            const float ka = (s < s1) ? elu1(a[u]) * m : 0.f;
            const float vb = (b[u] * m) / fS;                       // values / v_length (linear_attention.py:43)
            ks += ka;
            acc = M::mma(ka, vb, acc);
        }
    }
    // ksum: add the token-parity classes held by the different lane groups
#pragma unroll
    for (int d = D; d < 64; d <<= 1) ks += shfl_xor_f(ks, d);
    float* o = part + (((size_t)n * nchunk + ch) * HD + h * D) * (D + 1);
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) o[(size_t)M::row(r, kk) * (D + 1) + col] = acc[r];
    if (kk == 0) o[(size_t)col * (D + 1) + D] = ks;
}

// kv[n][HD][D+1] = sum over chunks (fixed order).
__global__ void k_la_kv_reduce(const float* __restrict__ part, int nchunk, int per_n, float* __restrict__ kv) {
    const int bpn = (per_n + blockDim.x - 1) / blockDim.x;
    int n = blockIdx.x / bpn, e = (blockIdx.x - n * bpn) * blockDim.x + threadIdx.x;
    if (e >= per_n) return;
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += part[((size_t)n * nchunk + c) * per_n + e];
    kv[(size_t)n * per_n + e] = s;
}

// out[l][h,v] = (sum_d Q'[l][h,d] KV[h][d][v]) / (sum_d Q'[l][h,d] ksum[h][d] + eps) * S.
// One wave per (n, head, block of token tiles): D[m = token][n = v]; A = Q' loaded as contiguous channel runs per
// lane, B = the head's KV columns held in registers for the whole block; the normaliser comes from a second MFMA
// against a B whose columns all equal ksum, so it lands in the same (token, v) register layout as the numerator.
template <int D>
__global__ __launch_bounds__(256) void k_la_apply(const float* __restrict__ q, const float* __restrict__ kv,
                                                  const uint8_t* __restrict__ q_mask, int N, int L, int S, int H,
                                                  int tiles_per_unit, float eps, float* __restrict__ out) {
    typedef Mf<D> M;
    constexpr int TT = D;                     // tokens per tile (32 or 16)
    constexpr int NK = D / M::KPER;           // MFMA steps per tile (16 or 4) = channels per lane
    const int lane = threadIdx.x & 63, col = lane & (D - 1), kk = lane >> M::SHIFT;
    const int ntile = (L + TT - 1) / TT, nblk = (ntile + tiles_per_unit - 1) / tiles_per_unit;
    const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= (long)N * nblk * H) return;
    const int h = (int)(unit % H);
    const int tb = (int)((unit / H) % nblk);
    const int n = (int)(unit / ((long)H * nblk));
    const int HD = H * D;
    // B operands: this lane's channels are d = NK*kk + t, t = 0..NK-1
    float bkv[NK], bks[NK];
    const float* kvn = kv + ((size_t)n * HD + h * D) * (D + 1);
#pragma unroll
    for (int t = 0; t < NK; ++t) {
        bkv[t] = kvn[(size_t)(NK * kk + t) * (D + 1) + col];
        bks[t] = kvn[(size_t)(NK * kk + t) * (D + 1) + D];
    }
    const float fS = (float)S;
    const int t_end = min(ntile, (tb + 1) * tiles_per_unit);
    // the token rows of tile ti + 1 are requested before tile ti is computed (raw values; elu + mask applied at use)
    float4 raw[NK / 4], nxt[NK / 4];
    float mk = 0.f, mk_n = 0.f;
    auto fetch = [&](int ti, float4 (&dst)[NK / 4], float& m) {
        const int l = ti * TT + col;          // the token this lane feeds into the A operand
        m = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < NK / 4; ++t4) dst[t4] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ti < t_end && l < L) {
            const float* qp = q + ((size_t)n * L + l) * HD + h * D + NK * kk;
            m = 1.f;
            if (q_mask) m = q_mask[(size_t)n * L + l] ? 1.f : 0.f;
#pragma unroll
            for (int t4 = 0; t4 < NK / 4; ++t4) dst[t4] = *reinterpret_cast<const float4*>(qp + 4 * t4);
        }
    };
    fetch(tb * tiles_per_unit, raw, mk);
    for (int ti = tb * tiles_per_unit; ti < t_end; ++ti) {
        fetch(ti + 1, nxt, mk_n);
        float qa[NK];
#pragma unroll
        for (int t4 = 0; t4 < NK / 4; ++t4) {
            // a lane outside the sequence contributes zeros (elu(0) + 1 = 1 would not): mk = 0 there
            qa[4 * t4 + 0] = elu1(raw[t4].x) * mk; qa[4 * t4 + 1] = elu1(raw[t4].y) * mk;
            qa[4 * t4 + 2] = elu1(raw[t4].z) * mk; qa[4 * t4 + 3] = elu1(raw[t4].w) * mk;
        }
        typename M::acc_t num, den;
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) { num[r] = 0.f; den[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            num = M::mma(qa[t], bkv[t], num);
            den = M::mma(qa[t], bks[t], den);
        }
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) {
            const int lo = ti * TT + M::row(r, kk);
            if (lo < L) {
                const float zz = 1.0f / (den[r] + eps);                              // linear_attention.py:46
                out[((size_t)n * L + lo) * HD + h * D + col] = (num[r] * zz) * fS;   // :50
            }
        }
#pragma unroll
        for (int t4 = 0; t4 < NK / 4; ++t4) raw[t4] = nxt[t4];
        mk = mk_n;
    }
}

template <int D>
int launch_la(const float* q, const float* k, const float* v, int N, int L, int S, int H, const uint8_t* q_mask,
              const uint8_t* kv_mask, float eps, float* out, float* ws, hipStream_t stream) {
    const int HD = H * D;
    int tok_per_chunk = S >= 1024 ? 320 : S;
    int nchunk = (S + tok_per_chunk - 1) / tok_per_chunk;
    size_t per_n = (size_t)HD * (D + 1);
    float* part = ws;
    float* kv = ws + (size_t)N * nchunk * per_n;
    long units = (long)N * nchunk * H;
    // a single chunk (short sequences, e.g. the 25-token fine windows) IS the reduced result: write it in place
    hipLaunchKernelGGL(k_la_kv_partial<D>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, k, v, kv_mask, N, S, H,
                       tok_per_chunk, nchunk, nchunk == 1 ? kv : part);
    if (nchunk > 1)
        hipLaunchKernelGGL(k_la_kv_reduce, dim3((int)((per_n + 255) / 256) * N), dim3(256), 0, stream, part, nchunk,
                           (int)per_n, kv);
    const int ntile = (L + D - 1) / D;
    int tiles_per_unit = ntile >= 64 ? 8 : ntile;
    int nblk = (ntile + tiles_per_unit - 1) / tiles_per_unit;
    long aunits = (long)N * nblk * H;
    hipLaunchKernelGGL(k_la_apply<D>, dim3((unsigned)((aunits + 3) / 4)), dim3(256), 0, stream, q, kv, q_mask, N, L, S, H,
                       tiles_per_unit, eps, out);
    return far_check_launch();
}

}  // namespace

extern "C" {

size_t far_linear_attention_workspace_bytes(int N, int S, int H, int D) {
    int tok_per_chunk = S >= 1024 ? 320 : S;
    int nchunk = (S + tok_per_chunk - 1) / tok_per_chunk;
    return ((size_t)N * nchunk + N) * (size_t)H * D * (D + 1) * sizeof(float);
}

// out [N][L][H*D] = LinearAttention(q [N][L][H*D], k, v [N][S][H*D]); D in {16, 32}; H*D in {128, 256}.
int far_linear_attention_f32(const float* q, const float* k, const float* v, int N, int L, int S, int H, int D,
                             const uint8_t* q_mask, const uint8_t* kv_mask, float eps, float* out, void* ws,
                             hipStream_t stream) {
    far_clear_errors();
    if (N == 0) return FAR_OK;
    if (!q || !k || !v || !out || !ws || N < 0 || L <= 0 || S <= 0) return FAR_EINVAL;
    const int HD = H * D;
    if (HD > 1024 || (HD % 64) != 0) return FAR_EINVAL;
    if (D == 32) return launch_la<32>(q, k, v, N, L, S, H, q_mask, kv_mask, eps, out, (float*)ws, stream);
    if (D == 16) return launch_la<16>(q, k, v, N, L, S, H, q_mask, kv_mask, eps, out, (float*)ws, stream);
    return FAR_EINVAL;
}

}  // extern "C"
