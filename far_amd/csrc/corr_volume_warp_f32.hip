// K12: the correlation-volume warp of the Map-free 6DReg aggregator.
//
// Replaces mapfree_6dreg/lib/models/regression/aggregator.py:44-115 (CorrelationVolumeWarping.forward) in the FAR
// configuration (config/regression/mapfree/rot6d_trans_with_loftr.yaml: POSITION_ENCODER, MAX_SCORE_CHANNEL; no dustbin,
// no normalisation, no half channels; 32 feature channels on a 92 x 68 grid):
//     cvolume  = softmax_j(vol0_i . vol1_j)                      (B, HW, HW)  = 156 MB per pair at HW = 6256, fp32
//     vol1w    = vol1 cvolume^T                                   (B, 32, HW)
//     pos_enc  = grid cvolume^T                                   (B, 2, HW)   grid = meshgrid(linspace(-1,1,H), linspace(-1,1,W))
//     max_sc   = max_j cvolume_ij                                 (B, 1, HW)
//     agg      = cat[vol0, vol1w, pos_enc, max_sc]                (B, 67, H, W)
// A single softmax (rows only), so it is a two-pass "attention" with q = vol0, k = v = vol1, never materialising
// cvolume.  Arithmetic: EXACT fp32 on the f32-input matrix core (v_mfma_f32_32x32x2_f32 is bitwise an fmaf chain): with
// K = 32 channels the contractions are 5 GFLOP per pair -- 40 us at the f32 MFMA rate -- so there is nothing to gain
// from split-fp16 operands here, and nothing to argue about parity.
//   pass 1 (k_cvw_stats)  row maximum and sum of exp over all columns (online, lane = row in the transposed score tile)
//   pass 2 (k_cvw_apply)  p = exp(x - max) / sum;  vol1w^T[ch][row] += vol1[ch][col] p[row][col] on the matrix core
//                         (D[m = channel][n = row]: a lane owns a row, so the 32 lanes of a half-wave store 128 contiguous
//                         bytes of one output channel); the two grid channels on the VALU (2 FMAs per score);
//                         max_sc = 1 / sum.
// The channel-major layout of the reference ((B, D, HW): position contiguous) is exactly what the MFMA operands want:
// A[k = channel][col] and B[k = channel][row] are 128-byte contiguous global reads; the column tile is staged once per
// workgroup in LDS ([32 ch][33] floats, conflict-free both as score operand and as warp operand).
#include "common.h"

namespace {

constexpr int D = 32;                 // feature channels (ENCODER.NUM_OUT_LAYERS)
constexpr int KT = 32;                // columns per tile
constexpr int LROW = KT + 1;          // padded LDS row (floats)
constexpr float LOG2E = 1.44269504088896341f;
constexpr float NEG_HUGE = -1.0e30f;

// stage vol1[b][0..31][j0 .. j0+31] -> lds[ch][LROW] (zeros past HW), grid[0..1][j0..] -> gl[2][KT]
__device__ __forceinline__ void stage_tile(float* lds, float* gl, const float* __restrict__ v1, const float* __restrict__ grid,
                                           int HW, int j0, int tid) {
    for (int e = tid; e < D * KT; e += 256) {
        const int ch = e >> 5, c = e & 31;
        lds[ch * LROW + c] = j0 + c < HW ? v1[(size_t)ch * HW + j0 + c] : 0.f;
    }
    if (tid < 2 * KT) {
        const int g = tid >> 5, c = tid & 31;
        gl[g * KT + c] = j0 + c < HW ? grid[(size_t)g * HW + j0 + c] : 0.f;
    }
}

// scores of one 32 x 32 tile, transposed: D[m = column][n = this lane's row]
__device__ __forceinline__ void score_tile(f32x16& sc, const float* lds, const float (&q)[16], int l31, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(2 * t + h) * LROW + l31], q[t], sc, 0, 0, 0);
}

__global__ __launch_bounds__(256) void k_cvw_stats(const float* __restrict__ vol0, const float* __restrict__ vol1, int B, int HW,
                                                   float2* __restrict__ stat) {
    __shared__ float lds[D * LROW];
    __shared__ float gl[2 * KT];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nI = (HW + 127) / 128;
    const int b = blockIdx.x / nI, Ib = blockIdx.x - b * nI;
    const int row = Ib * 128 + 32 * wave + l31;
    const float* v0 = vol0 + (size_t)b * D * HW;
    const float* v1 = vol1 + (size_t)b * D * HW;
    float q[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) q[t] = row < HW ? v0[(size_t)(2 * t + h) * HW + row] : 0.f;
    float m = NEG_HUGE, s = 0.f;
    for (int j0 = 0; j0 < HW; j0 += KT) {
        __syncthreads();
        stage_tile(lds, gl, v1, vol1, HW, j0, tid);          // (grid staging unused here: harmless reads of vol1)
        __syncthreads();
        f32x16 sc;
        score_tile(sc, lds, q, l31, h);
        float tm = NEG_HUGE;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = sc[r] * LOG2E;
            if (j0 + mfma32_row(r, h) >= HW) x = NEG_HUGE;
            sc[r] = x;
            tm = fmaxf(tm, x);
        }
        const float mn = fmaxf(m, tm);
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += __builtin_amdgcn_exp2f(sc[r] - mn);
        s = s * __builtin_amdgcn_exp2f(m - mn) + t;
        m = mn;
    }
    const float mo = shfl_xor_f(m, 32), so = shfl_xor_f(s, 32);
    const float mn = fmaxf(m, mo);
    const float st = s * __builtin_amdgcn_exp2f(m - mn) + so * __builtin_amdgcn_exp2f(mo - mn);
    if (h == 0 && row < HW) stat[(size_t)b * HW + row] = make_float2(mn, st);
}

__global__ __launch_bounds__(256) void k_cvw_apply(const float* __restrict__ vol0, const float* __restrict__ vol1,
                                                   const float* __restrict__ grid, const float2* __restrict__ stat, int B, int HW,
                                                   float* __restrict__ agg) {
    __shared__ float lds[D * LROW];
    __shared__ float gl[2 * KT];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int nI = (HW + 127) / 128;
    const int b = blockIdx.x / nI, Ib = blockIdx.x - b * nI;
    const int row = Ib * 128 + 32 * wave + l31;
    const float* v0 = vol0 + (size_t)b * D * HW;
    const float* v1 = vol1 + (size_t)b * D * HW;
    float q[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) q[t] = row < HW ? v0[(size_t)(2 * t + h) * HW + row] : 0.f;
    const float2 st = row < HW ? stat[(size_t)b * HW + row] : make_float2(0.f, 1.f);
    const float rinv = 1.0f / st.y;
    f32x16 acc;                                          // D[m = channel][n = row]
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float pu = 0.f, pv = 0.f;
    for (int j0 = 0; j0 < HW; j0 += KT) {
        __syncthreads();
        stage_tile(lds, gl, v1, grid, HW, j0, tid);
        __syncthreads();
        f32x16 sc;
        score_tile(sc, lds, q, l31, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = mfma32_row(r, h);
            float p = __builtin_amdgcn_exp2f(sc[r] * LOG2E - st.x) * rinv;
            if (j0 + c >= HW) p = 0.f;
            sc[r] = p;
            pu = fmaf(p, gl[c], pu);
            pv = fmaf(p, gl[KT + c], pv);
        }
        // vol1w^T[ch][row] += sum_col vol1[ch][col] p[row][col]: k-slot h of step r = column mfma32_row(r, h)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[l31 * LROW + mfma32_row(r, h)], sc[r], acc, 0, 0, 0);
    }
    pu += shfl_xor_f(pu, 32);
    pv += shfl_xor_f(pv, 32);
    if (row < HW) {
        float* o = agg + (size_t)b * (2 * D + 3) * HW + row;
        // lane (row, h) holds channels mfma32_row(r, h): per register 32 consecutive rows of one channel = 128 B per half-wave
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(size_t)(D + mfma32_row(r, h)) * HW] = acc[r];
#pragma unroll
        for (int t = 0; t < 16; ++t) o[(size_t)(2 * t + h) * HW] = q[t];           // the vol0 block of the concatenation
        if (h == 0) {
            o[(size_t)(2 * D) * HW] = pu;
            o[(size_t)(2 * D + 1) * HW] = pv;
            o[(size_t)(2 * D + 2) * HW] = rinv;                                     // max_j softmax = 2^(max - max) / sum
        }
    }
}

}  // namespace

extern "C" {

size_t far_corr_volume_warp_workspace_bytes(int B, int HW) { return B > 0 && HW > 0 ? (size_t)B * HW * sizeof(float2) : 0; }

// agg [B][2 D + 3][HW] = cat[vol0, vol1 softmax(vol0^T vol1)^T, grid softmax(..)^T, rowmax softmax(..)]  (D = 32)
// vol0, vol1 [B][D][HW] fp32 (the reference's (B, D, H, W) tensors), grid [2][HW].
int far_corr_volume_warp_f32(const float* vol0, const float* vol1, const float* grid, int B, int Dch, int HW, float* agg, void* ws,
                             hipStream_t stream) {
    far_clear_errors();
    if (!vol0 || !vol1 || !grid || !agg || !ws || B <= 0 || HW <= 0 || Dch != D) return FAR_EINVAL;
    const int nI = (HW + 127) / 128;
    hipLaunchKernelGGL(k_cvw_stats, dim3(nI * B), dim3(256), 0, stream, vol0, vol1, B, HW, (float2*)ws);
    hipLaunchKernelGGL(k_cvw_apply, dim3(nI * B), dim3(256), 0, stream, vol0, vol1, grid, (const float2*)ws, B, HW, agg);
    return far_check_launch();
}

}  // extern "C"
