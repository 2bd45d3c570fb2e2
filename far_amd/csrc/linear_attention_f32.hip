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
#include <algorithm>

namespace {

// tokens per KV partial sum of the forward (long sequences); knob 6 of far_set_tuning overrides it for A/B runs (the
// workspace query reads the same knob: set it before sizing the workspace)
static int fwd_chunk() { const int t = far_get_tuning(6); return t > 0 ? t : 320; }

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

// kv [.][D + 1] rows: multiply the D x D part by f, leave the last (ksum) column (backward: undoes the forward's 1/S)
__global__ void k_la_scale_kv(float* __restrict__ kv, long total, int D, float f) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total && (e % (D + 1)) != D) kv[e] *= f;
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
// lane, B = the head's KV columns held in registers for the whole block; the normaliser is a dot product over the channels a
// token's lanes already hold (fma chain + cross-lane adds).
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
        // transposed product D[m = v][n = token]: the lane that fed token `col` gets that token's outputs for the channels
        // v = row(r, kk) -- groups of four consecutive channels -> 16-byte stores (32 contiguous bytes per token and
        // instruction with the kk twin) instead of 4-byte ones
        typename M::acc_t num;
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) num[r] = 0.f;
        // the normaliser Q'[l] . ksum is a dot product of what this token's lanes already hold (channels NK kk + t): an fma
        // chain and one or two cross-lane adds instead of a second MFMA chain (half of this kernel's matrix work)
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            num = M::mma(bkv[t], qa[t], num);
            den = fmaf(qa[t], bks[t], den);
        }
#pragma unroll
        for (int m = D; m < 64; m <<= 1) den += shfl_xor_f(den, m);
        const int lo = ti * TT + col;
        if (lo < L) {
            const float zz = 1.0f / (den + eps);                                     // linear_attention.py:46
            float* op = out + ((size_t)n * L + lo) * HD + h * D;
#pragma unroll
            for (int g = 0; g < M::NACC / 4; ++g)                                    // :50
                *reinterpret_cast<float4*>(op + M::row(4 * g, kk)) = make_float4((num[4 * g] * zz) * fS, (num[4 * g + 1] * zz) * fS,
                                                                                 (num[4 * g + 2] * zz) * fS, (num[4 * g + 3] * zz) * fS);
        }
#pragma unroll
        for (int t4 = 0; t4 < NK / 4; ++t4) raw[t4] = nxt[t4];
        mk = mk_n;
    }
}

// Short sequences with 16-channel heads (the 25-token windows of the fine-level transformer: 61 k windows per 32 pairs):
// one wave per (window, head) does both steps.  KV (16 x 16) never leaves the accumulator registers -- the layout the
// first product leaves it in (lane (v, kk) holds KV[4 kk + r][v]) is the A operand of the second, transposed product
// with the channel order d = 4 kk + t -- so the 532 MB kv round trip of the two-kernel form (a third of its traffic)
// disappears.  Same products in the same order as k_la_kv_partial / k_la_apply.  No masks (the generic path has them).
__global__ __launch_bounds__(256) void k_la_window16(const float* __restrict__ q, const float* __restrict__ k,
                                                     const float* __restrict__ v, long units, int L, int S, int H, float eps,
                                                     float* __restrict__ out) {
    typedef Mf<16> M;
    const int lane = threadIdx.x & 63, col = lane & 15, kk = lane >> 4;
    const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= units) return;
    const int h = (int)(unit % H);
    const long n = unit / H;
    const int HD = H * 16;
    const float fS = (float)S;
    const float* kp = k + ((size_t)n * S) * HD + h * 16 + col;
    const float* vp = v + ((size_t)n * S) * HD + h * 16 + col;
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int s = 4 * u + kk;
        a[u] = 0.f; b[u] = 0.f;
        if (s < S) { a[u] = kp[(size_t)s * HD]; b[u] = vp[(size_t)s * HD]; }
    }
    float4 qv[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int l = 16 * ti + col;
        qv[ti] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < L) qv[ti] = *reinterpret_cast<const float4*>(q + ((size_t)n * L + l) * HD + h * 16 + 4 * kk);
    }
    f32x4 kv;
#pragma unroll
    for (int r = 0; r < 4; ++r) kv[r] = 0.f;
    float ks = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int s = 4 * u + kk;
        const float ka = s < S ? elu1(a[u]) : 0.f;
        ks += ka;
        kv = M::mma(ka, b[u] / fS, kv);                                   // values / v_length (linear_attention.py:43)
    }
    ks += shfl_xor_f(ks, 16);
    ks += shfl_xor_f(ks, 32);                                             // every lane with col = d holds ksum[d]
    float bks[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bks[t] = __shfl(ks, 4 * kk + t, 64);      // the normaliser's A operand: ksum[4 kk + t] in every row
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int l = 16 * ti + col;
        if (16 * ti >= L) break;
        const float mk = l < L ? 1.f : 0.f;
        const float qa[4] = {elu1(qv[ti].x) * mk, elu1(qv[ti].y) * mk, elu1(qv[ti].z) * mk, elu1(qv[ti].w) * mk};
        f32x4 num, den;
#pragma unroll
        for (int r = 0; r < 4; ++r) { num[r] = 0.f; den[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            num = M::mma(kv[t], qa[t], num);
            den = M::mma(bks[t], qa[t], den);
        }
        if (l < L) {
            const float zz = 1.0f / (den[0] + eps);
            *reinterpret_cast<float4*>(out + ((size_t)n * L + l) * HD + h * 16 + 4 * kk) =
                make_float4((num[0] * zz) * fS, (num[1] * zz) * fS, (num[2] * zz) * fS, (num[3] * zz) * fS);
        }
    }
}

template <int D>
int launch_la(const float* q, const float* k, const float* v, int N, int L, int S, int H, const uint8_t* q_mask,
              const uint8_t* kv_mask, float eps, float* out, float* ws, hipStream_t stream) {
    const int HD = H * D;
    if (D == 16 && L <= 32 && S <= 32 && !q_mask && !kv_mask && far_get_tuning(4) == 0) {
        const long units = (long)N * H;
        hipLaunchKernelGGL(k_la_window16, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, q, k, v, units, L, S, H, eps, out);
        return far_check_launch();
    }
    int tok_per_chunk = S >= 1024 ? fwd_chunk() : S;
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
    int tiles_per_unit = ntile >= 64 ? 4 : ntile;     // measured at 32 x 4800 x 256: 2-4 tiles 169-172 us (kv + apply), 8: 179, 16: 186
    if (far_get_tuning(5) > 0) tiles_per_unit = std::min(ntile, far_get_tuning(5));      // A/B knob
    int nblk = (ntile + tiles_per_unit - 1) / tiles_per_unit;
    long aunits = (long)N * nblk * H;
    hipLaunchKernelGGL(k_la_apply<D>, dim3((unsigned)((aunits + 3) / 4)), dim3(256), 0, stream, q, kv, q_mask, N, L, S, H,
                       tiles_per_unit, eps, out);
    return far_check_launch();
}

}  // namespace

extern "C" {

size_t far_linear_attention_workspace_bytes(int N, int S, int H, int D) {
    int tok_per_chunk = S >= 1024 ? fwd_chunk() : S;
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

// The second half alone: out [N][L][H*32] from q [N][L][H*32] and the state kv [N][H*32][33] (K'^T (V / S) and, in the 33rd
// column, the sum of K') that far_linear_kv_f16s leaves -- the same k_la_apply launch far_linear_attention_f32 ends with.
int far_linear_attention_apply_f32(const float* q, const float* kv, int N, int L, int S, int H, const uint8_t* q_mask, float eps,
                                   float* out, hipStream_t stream) {
    far_clear_errors();
    if (N == 0) return FAR_OK;
    if (!q || !kv || !out || N < 0 || L <= 0 || S <= 0 || H <= 0 || H * 32 > 1024) return FAR_EINVAL;
    const int ntile = (L + 31) / 32;
    int tiles_per_unit = ntile >= 64 ? 4 : ntile;
    if (far_get_tuning(5) > 0) tiles_per_unit = std::min(ntile, far_get_tuning(5));
    const int nblk = (ntile + tiles_per_unit - 1) / tiles_per_unit;
    const long aunits = (long)N * nblk * H;
    hipLaunchKernelGGL(k_la_apply<32>, dim3((unsigned)((aunits + 3) / 4)), dim3(256), 0, stream, q, kv, q_mask, N, L, S, H,
                       tiles_per_unit, eps, out);
    return far_check_launch();
}

}  // extern "C"

// =====================================================================================================================
// K5 backward (training path, BASELINE configs[2]): gradients of LinearAttention.forward w.r.t. the raw projections.
// With Q' = elu(q) + 1, K' = elu(k) + 1 (masked), KV = K'^T V, ksum = sum_s K'_s, den_l = Q'_l . ksum + eps,
// out_l = Q'_l KV / den_l  (the 1/S and *S of linear_attention.py:43,50 cancel), and g = dL/dout:
//   dnum_l = g_l / den_l,   dden_l = -(g_l . out_l) / den_l
//   dQ'_l  = dnum_l KV^T + dden_l ksum                        dq = dQ' * elu'(q)   (elu'(x) = 1 for x > 0, e^x otherwise)
//   dKV    = sum_l Q'_l^T dnum_l,   dksum = sum_l dden_l Q'_l (the forward's K^T V kernel with a per-token weight)
//   dK'_s  = dKV V_s + dksum,  dV_s = dKV^T K'_s              dk = dK' * elu'(k)
// Token-parallel kernels (one thread = one token of one head; the head's D x D matrices in LDS), bandwidth-bound like
// the forward; the token reductions reuse k_la_kv_partial / k_la_kv_reduce (fixed order: deterministic).
// =====================================================================================================================
namespace {

__device__ __forceinline__ float elu1_grad(float x) { return x > 0.f ? 1.f : expf(x); }

// per (n, head, token l): recompute num / den, write dq, dnum (workspace [N][L][HD]) and dden (workspace [N][L][H])
template <int D>
__global__ __launch_bounds__(256) void k_la_bwd_q(const float* __restrict__ q, const float* __restrict__ g,
                                                  const float* __restrict__ kv, const uint8_t* __restrict__ q_mask, int N, int L,
                                                  int H, float eps, float* __restrict__ dq, float* __restrict__ dnum,
                                                  float* __restrict__ dden) {
    __shared__ float s_kv[D][D + 1];
    __shared__ float s_ks[D];
    const int h = blockIdx.y, n = blockIdx.z, HD = H * D;
    const float* kvn = kv + ((size_t)n * HD + h * D) * (D + 1);
    for (int e = threadIdx.x; e < D * (D + 1); e += blockDim.x) {
        const int d = e / (D + 1), c = e - d * (D + 1);
        if (c < D) s_kv[d][c] = kvn[e]; else s_ks[d] = kvn[e];
    }
    __syncthreads();
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const size_t off = ((size_t)n * L + l) * HD + h * D;
    const float m = q_mask ? (q_mask[(size_t)n * L + l] ? 1.f : 0.f) : 1.f;
    float qr[D], Q[D], G[D];
#pragma unroll
    for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 a = *reinterpret_cast<const float4*>(q + off + 4 * d4), b = *reinterpret_cast<const float4*>(g + off + 4 * d4);
        qr[4 * d4] = a.x; qr[4 * d4 + 1] = a.y; qr[4 * d4 + 2] = a.z; qr[4 * d4 + 3] = a.w;
        G[4 * d4] = b.x; G[4 * d4 + 1] = b.y; G[4 * d4 + 2] = b.z; G[4 * d4 + 3] = b.w;
    }
    float den = eps;
#pragma unroll
    for (int d = 0; d < D; ++d) { Q[d] = elu1(qr[d]) * m; den += Q[d] * s_ks[d]; }
    const float rden = 1.0f / den;
    float go = 0.f;                                        // g . num
    float dn[D];
#pragma unroll
    for (int v = 0; v < D; ++v) {
        float num = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) num += Q[d] * s_kv[d][v];
        go += G[v] * num;
        dn[v] = G[v] * rden;
    }
    const float dd = -(go * rden) * rden;                  // -(g . out) / den
    float o4[4];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float a = dd * s_ks[d];
#pragma unroll
        for (int v = 0; v < D; ++v) a += dn[v] * s_kv[d][v];
        o4[d & 3] = a * m * elu1_grad(qr[d]);
        if ((d & 3) == 3) *reinterpret_cast<float4*>(dq + off + d - 3) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
#pragma unroll
    for (int d4 = 0; d4 < D / 4; ++d4)
        *reinterpret_cast<float4*>(dnum + off + 4 * d4) = make_float4(dn[4 * d4], dn[4 * d4 + 1], dn[4 * d4 + 2], dn[4 * d4 + 3]);
    dden[((size_t)n * L + l) * H + h] = dd;
}

// dKV / dksum partials: the forward's K^T V accumulation with (K -> Q', V -> dnum) and ksum weighted by dden per token
template <int D>
__global__ __launch_bounds__(256) void k_la_bwd_dkv_partial(const float* __restrict__ q, const float* __restrict__ dnum,
                                                            const float* __restrict__ dden, const uint8_t* __restrict__ q_mask,
                                                            int N, int L, int H, int tok_per_chunk, int nchunk,
                                                            float* __restrict__ part) {
    typedef Mf<D> M;
    const int lane = threadIdx.x & 63, col = lane & (D - 1), kk = lane >> M::SHIFT;
    const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= (long)N * nchunk * H) return;
    const int h = (int)(unit % H);
    const int ch = (int)((unit / H) % nchunk);
    const int n = (int)(unit / ((long)H * nchunk));
    const int HD = H * D;
    const int s0 = ch * tok_per_chunk, s1 = min(L, s0 + tok_per_chunk);
    typename M::acc_t acc;
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) acc[r] = 0.f;
    float ks = 0.f;
    for (int sb = s0; sb < s1; sb += M::KPER) {
        const int s = sb + kk;
        float a = 0.f, b = 0.f, w = 0.f;
        if (s < s1) {
            const size_t o = ((size_t)n * L + s) * HD + h * D + col;
            const float m = q_mask ? (q_mask[(size_t)n * L + s] ? 1.f : 0.f) : 1.f;
            a = elu1(q[o]) * m;
            b = dnum[o];
            w = dden[((size_t)n * L + s) * H + h];
        }
        ks += a * w;
        acc = M::mma(a, b, acc);
    }
#pragma unroll
    for (int d = D; d < 64; d <<= 1) ks += shfl_xor_f(ks, d);
    float* o = part + (((size_t)n * nchunk + ch) * HD + h * D) * (D + 1);
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) o[(size_t)M::row(r, kk) * (D + 1) + col] = acc[r];
    if (kk == 0) o[(size_t)col * (D + 1) + D] = ks;
}

// per (n, head, token s): dk = (dKV V_s + dksum) * elu'(k) * mask,  dv = dKV^T K'_s * mask
template <int D>
__global__ __launch_bounds__(256) void k_la_bwd_kv(const float* __restrict__ k, const float* __restrict__ v,
                                                   const float* __restrict__ dkv, const uint8_t* __restrict__ kv_mask, int N, int S,
                                                   int H, float* __restrict__ dk, float* __restrict__ dv) {
    __shared__ float s_kv[D][D + 1];
    __shared__ float s_ks[D];
    const int h = blockIdx.y, n = blockIdx.z, HD = H * D;
    const float* kvn = dkv + ((size_t)n * HD + h * D) * (D + 1);
    for (int e = threadIdx.x; e < D * (D + 1); e += blockDim.x) {
        const int d = e / (D + 1), c = e - d * (D + 1);
        if (c < D) s_kv[d][c] = kvn[e]; else s_ks[d] = kvn[e];
    }
    __syncthreads();
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const size_t off = ((size_t)n * S + s) * HD + h * D;
    const float m = kv_mask ? (kv_mask[(size_t)n * S + s] ? 1.f : 0.f) : 1.f;
    float kr[D], V[D];
#pragma unroll
    for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 a = *reinterpret_cast<const float4*>(k + off + 4 * d4), b = *reinterpret_cast<const float4*>(v + off + 4 * d4);
        kr[4 * d4] = a.x; kr[4 * d4 + 1] = a.y; kr[4 * d4 + 2] = a.z; kr[4 * d4 + 3] = a.w;
        V[4 * d4] = b.x * m; V[4 * d4 + 1] = b.y * m; V[4 * d4 + 2] = b.z * m; V[4 * d4 + 3] = b.w * m;
    }
    float o4[4];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float a = s_ks[d];
#pragma unroll
        for (int c = 0; c < D; ++c) a += s_kv[d][c] * V[c];
        o4[d & 3] = a * m * elu1_grad(kr[d]);
        if ((d & 3) == 3) *reinterpret_cast<float4*>(dk + off + d - 3) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
    float Kp[D];
#pragma unroll
    for (int d = 0; d < D; ++d) Kp[d] = elu1(kr[d]) * m;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) a += Kp[d] * s_kv[d][c];
        o4[c & 3] = a * m;
        if ((c & 3) == 3) *reinterpret_cast<float4*>(dv + off + c - 3) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
}

template <int D>
int launch_la_bwd(const float* q, const float* k, const float* v, const float* g, int N, int L, int S, int H,
                  const uint8_t* q_mask, const uint8_t* kv_mask, float eps, float* dq, float* dk, float* dv, float* ws,
                  hipStream_t stream) {
    const int HD = H * D;
    const size_t per_n = (size_t)HD * (D + 1);
    // 1. KV / ksum of the forward, recomputed WITHOUT the 1/S of values (it cancels against the final *S)
    int tpc = S >= 1024 ? 320 : S, nch = (S + tpc - 1) / tpc;
    int tpl = L >= 1024 ? 320 : L, ncl = (L + tpl - 1) / tpl;
    float* part = ws;                                                     // max(nch, ncl) * N * per_n
    float* kv = part + (size_t)N * std::max(nch, ncl) * per_n;            // N * per_n
    float* dkv = kv + (size_t)N * per_n;                                  // N * per_n
    float* dnum = dkv + (size_t)N * per_n;                                // N * L * HD
    float* dden = dnum + (size_t)N * L * HD;                              // N * L * H
    long units = (long)N * nch * H;
    hipLaunchKernelGGL(k_la_kv_partial<D>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, k, v, kv_mask, N, S, H, tpc, nch,
                       nch == 1 ? kv : part);
    if (nch > 1) hipLaunchKernelGGL(k_la_kv_reduce, dim3((int)((per_n + 255) / 256) * N), dim3(256), 0, stream, part, nch, (int)per_n, kv);
    // k_la_kv_partial divides the values by S: undo it on the D x D block (ksum column is unscaled)
    hipLaunchKernelGGL(k_la_scale_kv, dim3((unsigned)((N * per_n + 255) / 256)), dim3(256), 0, stream, kv, (long)N * per_n, D, (float)S);
    // 2. per query token: dq, dnum, dden
    hipLaunchKernelGGL(k_la_bwd_q<D>, dim3((L + 255) / 256, H, N), dim3(256), 0, stream, q, g, kv, q_mask, N, L, H, eps, dq, dnum, dden);
    // 3. dKV, dksum
    units = (long)N * ncl * H;
    hipLaunchKernelGGL(k_la_bwd_dkv_partial<D>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, q, dnum, dden, q_mask, N, L, H,
                       tpl, ncl, ncl == 1 ? dkv : part);
    if (ncl > 1) hipLaunchKernelGGL(k_la_kv_reduce, dim3((int)((per_n + 255) / 256) * N), dim3(256), 0, stream, part, ncl, (int)per_n, dkv);
    // 4. per key token: dk, dv
    hipLaunchKernelGGL(k_la_bwd_kv<D>, dim3((S + 255) / 256, H, N), dim3(256), 0, stream, k, v, dkv, kv_mask, N, S, H, dk, dv);
    return far_check_launch();
}

}  // namespace

extern "C" {

size_t far_linear_attention_bwd_workspace_bytes(int N, int L, int S, int H, int D) {
    const size_t per_n = (size_t)H * D * (D + 1);
    const int nch = (S + (S >= 1024 ? 320 : S) - 1) / (S >= 1024 ? 320 : S), ncl = (L + (L >= 1024 ? 320 : L) - 1) / (L >= 1024 ? 320 : L);
    return ((size_t)N * std::max(nch, ncl) * per_n + 2 * (size_t)N * per_n + (size_t)N * L * H * D + (size_t)N * L * H) * sizeof(float);
}

// Gradients of far_linear_attention_f32 w.r.t. q, k, v given g = dL/dout (all [N][tokens][H*D] fp32, contiguous).
int far_linear_attention_bwd_f32(const float* q, const float* k, const float* v, const float* g, int N, int L, int S, int H, int D,
                                 const uint8_t* q_mask, const uint8_t* kv_mask, float eps, float* dq, float* dk, float* dv,
                                 void* ws, hipStream_t stream) {
    far_clear_errors();
    if (N == 0) return FAR_OK;
    if (!q || !k || !v || !g || !dq || !dk || !dv || !ws || N < 0 || L <= 0 || S <= 0) return FAR_EINVAL;
    const int HD = H * D;
    if (HD > 1024 || (HD % 64) != 0) return FAR_EINVAL;
    if (D == 32) return launch_la_bwd<32>(q, k, v, g, N, L, S, H, q_mask, kv_mask, eps, dq, dk, dv, (float*)ws, stream);
    if (D == 16) return launch_la_bwd<16>(q, k, v, g, N, L, S, H, q_mask, kv_mask, eps, dq, dk, dv, (float*)ws, stream);
    return FAR_EINVAL;
}

}  // extern "C"
