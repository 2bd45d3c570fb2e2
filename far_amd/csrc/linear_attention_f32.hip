// K5: the linear-attention core of the LoFTR encoder layers.
//
// Replaces mp3d_loftr/src/loftr/loftr_module/linear_attention.py:31-50 (LinearAttention.forward):
//   Q = elu(q)+1, K = elu(k)+1; values = v / S
//   KV = einsum(nshd,nshv->nhdv, K, values);  Z = 1 / (einsum(nlhd,nhd->nlh, Q, K.sum(1)) + eps)
//   out = einsum(nlhd,nhdv,nlh->nlhv, Q, KV, Z) * S
// Inputs are the raw projections q [N][L][H*D], k, v [N][S][H*D]; the feature map (elu+1), the 1/S and *S
// scalings and the normaliser are fused.  Two kernels: a token-chunked reduction producing KV and K.sum
// (deterministic two-level sum, no atomics) and a streaming apply.
#include "common.h"

namespace {

__device__ __forceinline__ float elu1(float x) { return (x > 0.f ? x : expm1f(x)) + 1.f; }  // F.elu(x) + 1

constexpr int LA_TOK = 16;  // tokens staged per step

// grid (nchunk, N), block HD threads.  Thread t = (h, d) accumulates KV[h][d][0..D) and ksum[h][d] over its
// chunk of tokens.  part: [N][nchunk][HD][D+1]  (last column = ksum).
template <int D>
__global__ void k_la_kv_partial(const float* __restrict__ k, const float* __restrict__ v,
                                const uint8_t* __restrict__ kv_mask,  // optional [N][S]
                                int S, int HD, int tok_per_chunk, int nchunk, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* vs = sm;  // [LA_TOK][HD]
    const int t = threadIdx.x, n = blockIdx.x / nchunk, ch = blockIdx.x - n * nchunk;
    const int hbase = (t / D) * D;
    const int s0 = ch * tok_per_chunk, s1 = min(S, s0 + tok_per_chunk);
    float acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = 0.f;
    float ksum = 0.f;
    const float fS = (float)S;
    for (int sb = s0; sb < s1; sb += LA_TOK) {
        const int nt = min(LA_TOK, s1 - sb);
        float kf[LA_TOK];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LA_TOK; ++i) {
            float kk = 0.f, vv = 0.f;
            if (i < nt) {
                size_t off = ((size_t)n * S + sb + i) * HD + t;
                float m = (kv_mask && !kv_mask[(size_t)n * S + sb + i]) ? 0.f : 1.f;
                kk = elu1(k[off]) * m;
                vv = (v[off] * m) / fS;   // values / v_length (linear_attention.py:43)
            }
            kf[i] = kk;
            vs[i * HD + t] = vv;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LA_TOK; ++i) {
            const float kk = kf[i];
            ksum += kk;
            const float4* vr = reinterpret_cast<const float4*>(&vs[i * HD + hbase]);
#pragma unroll
            for (int c = 0; c < D / 4; ++c) {
                float4 x = vr[c];
                acc[4 * c + 0] = fmaf(kk, x.x, acc[4 * c + 0]);
                acc[4 * c + 1] = fmaf(kk, x.y, acc[4 * c + 1]);
                acc[4 * c + 2] = fmaf(kk, x.z, acc[4 * c + 2]);
                acc[4 * c + 3] = fmaf(kk, x.w, acc[4 * c + 3]);
            }
        }
    }
    float* o = part + (((size_t)n * nchunk + ch) * HD + t) * (D + 1);
#pragma unroll
    for (int i = 0; i < D; ++i) o[i] = acc[i];
    o[D] = ksum;
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

// grid (ceil(L / tok_per_block), N), block HD threads; thread (h, vch) keeps the KV column KV[h][:, vch]
// and ksum[h][:] in registers and streams tokens.
template <int D>
__global__ void k_la_apply(const float* __restrict__ q, const float* __restrict__ kv,
                           const uint8_t* __restrict__ q_mask,  // optional [N][L]
                           int L, int S, int HD, int tok_per_block, float eps, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm;  // [LA_TOK][HD]
    const int nblk = (L + tok_per_block - 1) / tok_per_block;
    const int t = threadIdx.x, n = blockIdx.x / nblk, bx = blockIdx.x - n * nblk;
    const int h = t / D, vch = t - h * D, hbase = h * D;
    float kvc[D], ks[D];
    const float* kvn = kv + (size_t)n * HD * (D + 1);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        kvc[d] = kvn[(size_t)(hbase + d) * (D + 1) + vch];
        ks[d] = kvn[(size_t)(hbase + d) * (D + 1) + D];
    }
    const int l0 = bx * tok_per_block, l1 = min(L, l0 + tok_per_block);
    const float fS = (float)S;
    for (int lb = l0; lb < l1; lb += LA_TOK) {
        const int nt = min(LA_TOK, l1 - lb);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LA_TOK; ++i) {
            float qq = 0.f;
            if (i < nt) {
                float m = (q_mask && !q_mask[(size_t)n * L + lb + i]) ? 0.f : 1.f;
                qq = elu1(q[((size_t)n * L + lb + i) * HD + t]) * m;
            }
            qs[i * HD + t] = qq;
        }
        __syncthreads();
        for (int i = 0; i < nt; ++i) {
            const float4* qr = reinterpret_cast<const float4*>(&qs[i * HD + hbase]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int c = 0; c < D / 4; ++c) {
                float4 x = qr[c];
                num = fmaf(x.x, kvc[4 * c + 0], num); den = fmaf(x.x, ks[4 * c + 0], den);
                num = fmaf(x.y, kvc[4 * c + 1], num); den = fmaf(x.y, ks[4 * c + 1], den);
                num = fmaf(x.z, kvc[4 * c + 2], num); den = fmaf(x.z, ks[4 * c + 2], den);
                num = fmaf(x.w, kvc[4 * c + 3], num); den = fmaf(x.w, ks[4 * c + 3], den);
            }
            float zz = 1.0f / (den + eps);                          // linear_attention.py:46
            out[((size_t)n * L + lb + i) * HD + t] = (num * zz) * fS;  // :50
        }
    }
}

template <int D>
int launch_la(const float* q, const float* k, const float* v, int N, int L, int S, int H, const uint8_t* q_mask,
              const uint8_t* kv_mask, float eps, float* out, float* ws, hipStream_t stream) {
    const int HD = H * D;
    // chunking: ~64 tokens per chunk keeps >= 64 blocks per image pair side at S = 4800
    int tok_per_chunk = S >= 1024 ? 64 : S;
    int nchunk = (S + tok_per_chunk - 1) / tok_per_chunk;
    size_t per_n = (size_t)HD * (D + 1);
    float* part = ws;
    float* kv = ws + (size_t)N * nchunk * per_n;
    size_t smem = (size_t)LA_TOK * HD * sizeof(float);
    hipLaunchKernelGGL(k_la_kv_partial<D>, dim3(nchunk * N), dim3(HD), smem, stream, k, v, kv_mask, S, HD,
                       tok_per_chunk, nchunk, part);
    hipLaunchKernelGGL(k_la_kv_reduce, dim3((int)((per_n + 255) / 256) * N), dim3(256), 0, stream, part, nchunk,
                       (int)per_n, kv);
    int tok_per_block = L >= 1024 ? 64 : L;
    hipLaunchKernelGGL(k_la_apply<D>, dim3(((L + tok_per_block - 1) / tok_per_block) * N), dim3(HD), smem, stream, q, kv,
                       q_mask, L, S, HD, tok_per_block, eps, out);
    return far_check_launch();
}

}  // namespace

extern "C" {

size_t far_linear_attention_workspace_bytes(int N, int S, int H, int D) {
    int tok_per_chunk = S >= 1024 ? 64 : S;
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
