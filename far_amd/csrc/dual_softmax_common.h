// Shared pieces of the dual-softmax kernels (K1 fp32, K1 bf16, K2 statistics): similarity parameters, the
// selection / compaction kernels that follow the match pass, and the workspace layout.
#pragma once
#include "gemm_tile_f32.h"

namespace far_ds {


constexpr float NEG_BIG = -FLT_MAX;

struct SimParams {
    float feat_div;   // features divided by this when staged; 1 when the scaling is folded into acc_scale
    float acc_scale;  // exact power-of-two factor applied to the dot product (1/feat_div^2 when folded, else 1)
    float sim_div;    // then / sim_div           (temperature for K1, 1 for K2)
    float sim_rcp;    // 1 / sim_div (IEEE), for the 3-instruction exact division
    float sim_mul;    // then * sim_mul           (1 for K1, head_dim^-0.5 for K2)
    float mask_fill;  // value for masked-out (i,j) pairs (-1e9 in the reference)
    int stagger;      // tuning: wave-slot priority staggering on/off
    float k2;         // acc_scale / sim_div * sim_mul * log2(e): log2-domain score per unit dot product (bf16 path)
};

// The reference divides both feature maps by sqrt(C) before the contraction (coarse_matching.py:104-105).  When
// sqrt(C) is a power of two that scaling commutes exactly with every rounding of the fmaf chain, so it is applied
// once to the accumulator instead of to 2 x 32 floats per thread per chunk: bit-identical, far fewer instructions.
inline SimParams make_sim(float feat_div, float sim_div, float sim_mul) {
    SimParams p;
    int e;
    float m = frexpf(feat_div, &e);
    bool pow2 = (m == 0.5f);
    p.feat_div = pow2 ? 1.0f : feat_div;
    p.acc_scale = pow2 ? 1.0f / (feat_div * feat_div) : 1.0f;
    p.sim_div = sim_div;
    p.sim_rcp = 1.0f / sim_div;
    p.sim_mul = sim_mul;
    p.mask_fill = -1e9f;
    p.stagger = 0;
    p.k2 = (float)((double)p.acc_scale / (double)sim_div * (double)sim_mul * 1.4426950408889634);
    return p;
}

__device__ __forceinline__ float sim_of(float acc, const SimParams& p) {
    float s = fdiv_by(acc * p.acc_scale, p.sim_div, p.sim_rcp);
    return s * p.sim_mul;
}

// --------------------------------------------------------------------------------------------
// finalize: match_j[z][i] = j* if row i is a mutual-nearest match above threshold and inside the
// border, else -1.  Border semantics of mask_border (coarse_matching.py:8-25): a cell (y,x) of an
// h x w grid survives iff bd <= y < h-bd and bd <= x < w-bd.  With padded masks
// (mask_border_with_padding, :28-43) the lower limits come from per-sample valid extents hv/wv.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ bool border_ok(int idx, int w, int hlim, int wlim, int bd) {
    int y = idx / w, x = idx - y * w;
    return y >= bd && y < hlim - bd && x >= bd && x < wlim - bd;
}

static __global__ void k_finalize(const float* __restrict__ rowbest_v, const int* __restrict__ rowbest_j,
                           const float* __restrict__ colbest_part, int nI, int L, int S, float thr,
                           int bd, int h0, int w0, int h1, int w1,
                           const int* __restrict__ valid_hw,  // optional [Z][4] = h0v,w0v,h1v,w1v
                           int* __restrict__ match_j, int* __restrict__ counts) {
    const int z = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int hl0 = h0, wl0 = w0, hl1 = h1, wl1 = w1;
    if (valid_hw) { hl0 = valid_hw[z * 4]; wl0 = valid_hw[z * 4 + 1]; hl1 = valid_hw[z * 4 + 2]; wl1 = valid_hw[z * 4 + 3]; }
    int mj = -1;
    if (i < L) {
        float v = rowbest_v[(size_t)z * L + i];
        int j = rowbest_j[(size_t)z * L + i];
        if (v > thr && j < S) {
            bool ok = bd <= 0 || (border_ok(i, w0, hl0, wl0, bd) && border_ok(j, w1, hl1, wl1, bd));
            if (ok) {
                float cm = -1.f;
                for (int b = 0; b < nI; ++b) cm = fmaxf(cm, colbest_part[((size_t)z * nI + b) * S + j]);
                if (v == cm) mj = j;
            }
        }
        match_j[(size_t)z * L + i] = mj;
    }
    unsigned long long bal = __ballot(mj >= 0);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&counts[z], __popcll(bal));
}

// compact: one block per pair; offset = sum of counts of earlier pairs; ordered by i.
static __global__ void k_compact(const int* __restrict__ match_j, const float* __restrict__ rowbest_v,
                          const int* __restrict__ counts, int L, int w0, int w1, float scale,
                          const float* __restrict__ scale0, const float* __restrict__ scale1,  // optional [Z][2]
                          int64_t* __restrict__ b_ids, int64_t* __restrict__ i_ids,
                          int64_t* __restrict__ j_ids, float* __restrict__ mconf,
                          float* __restrict__ mkpts0, float* __restrict__ mkpts1,
                          int* __restrict__ total) {
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int z = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        int off = 0;
        for (int b = 0; b < z; ++b) off += counts[b];
        base_s = off;
        if (z == gridDim.x - 1) *total = off + counts[z];
    }
    __syncthreads();
    int base = base_s;
    for (int i0 = 0; i0 < L; i0 += 256) {
        int i = i0 + tid;
        int mj = (i < L) ? match_j[(size_t)z * L + i] : -1;
        unsigned long long bal = __ballot(mj >= 0);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_cnt[w];
        int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (mj >= 0) {
            int pos = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
            b_ids[pos] = z;
            i_ids[pos] = i;
            j_ids[pos] = mj;
            mconf[pos] = rowbest_v[(size_t)z * L + i];
            float sx0 = scale, sy0 = scale, sx1 = scale, sy1 = scale;
            if (scale0) { sx0 = scale * scale0[z * 2]; sy0 = scale * scale0[z * 2 + 1]; }
            if (scale1) { sx1 = scale * scale1[z * 2]; sy1 = scale * scale1[z * 2 + 1]; }
            mkpts0[2 * pos] = (float)(i % w0) * sx0;
            mkpts0[2 * pos + 1] = (float)(i / w0) * sy0;
            mkpts1[2 * pos] = (float)(mj % w1) * sx1;
            mkpts1[2 * pos + 1] = (float)(mj / w1) * sy1;
        }
        base += tot;
        __syncthreads();
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct K1Workspace {
    float2* rowstat; float2* colpart; float2* colstat;
    float* rowbest_v; int* rowbest_j; float* colbest_part; int* match_j; int* counts; int* total;
    size_t bytes;
};

inline K1Workspace carve(void* ws, int Z, int L, int S) {
    K1Workspace w;
    int nI = (L + TILE_M - 1) / TILE_M;
    char* p = reinterpret_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off += align256(n); return q; };
    w.rowstat = (float2*)take((size_t)Z * L * sizeof(float2));
    w.colpart = (float2*)take((size_t)Z * nI * S * sizeof(float2));
    w.colstat = (float2*)take((size_t)Z * S * sizeof(float2));
    w.rowbest_v = (float*)take((size_t)Z * L * sizeof(float));
    w.rowbest_j = (int*)take((size_t)Z * L * sizeof(int));
    w.colbest_part = (float*)take((size_t)Z * nI * S * sizeof(float));
    w.match_j = (int*)take((size_t)Z * L * sizeof(int));
    w.counts = (int*)take((size_t)(Z + 1) * sizeof(int));
    w.total = w.counts ? w.counts + Z : nullptr;
    w.bytes = off;
    return w;
}

}  // namespace far_ds
