// K11: the small per-pair arithmetic between the solver and the regression head, one launch each instead of the
// 20-30 single-element torch kernels the same lines cost on the host-driven path (1.2 ms of idle GPU per solver round
// at batch 32, profiles/r01_bench_fp32_kernel_trace.txt):
//  * far_pose_pack_f64      mp3d_loftr/src/loftr/utils/supervision.py:218-233 (spvs_RT: [R | t] with the identity
//                           fallback of :221-224, E with its identity fallback, the count tensors) and
//                           src/utils/metrics.py:83-85 (fewer than 5 correspondences: all counts 0)
//  * far_pose_features_f32  mp3d_loftr/src/loftr/loftr.py:137-171 (preprocess_helper: the pose and its inverse as
//                           normalised [t, first two rows of R], plus correspondence counts / 500) with
//                           src/losses/loftr_loss.py:7-8, 31-39 (pose_mean_6d / pose_std_6d, compute_normalized_6d)
#include "common.h"

namespace {

// dataset statistics of the normalised 9-vector [t, R row 0, R row 1] (loftr_loss.py:7-8; far_amd/pose6d.py)
__constant__ float c_mean[9] = {-0.34898765f, 0.17085525f, -0.87944315f, 0.50275223f, 0.03533648f, -0.18179045f,
                                -0.03533648f, 0.98189617f, 0.09313615f};
__constant__ float c_std[9] = {1.94014405f, 0.36770130f, 1.88317520f, 0.51837117f, 0.12717603f, 0.65426397f,
                               0.12717603f, 0.0188729f, 0.09709263f};

__global__ void k_pose_pack(const double* __restrict__ R, const double* __restrict__ t, const double* __restrict__ E,
                            const int* __restrict__ status, const int* __restrict__ num_after,
                            const int* __restrict__ tight, const int* __restrict__ ultra, const int* __restrict__ offsets,
                            int B, double* __restrict__ rt_out, double* __restrict__ E_out, long* __restrict__ before_out,
                            int* __restrict__ after_out, int* __restrict__ tight_out, int* __restrict__ ultra_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool ok = status[b] != 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rt_out[b * 12 + 4 * r + c] = ok ? R[b * 9 + 3 * r + c] : (r == c ? 1.0 : 0.0);       // supervision.py:221-224
            E_out[b * 9 + 3 * r + c] = ok ? E[b * 9 + 3 * r + c] : (r == c ? 1.0 : 0.0);
        }
        rt_out[b * 12 + 4 * r + 3] = ok ? t[b * 3 + r] : 0.0;
    }
    const int cnt = offsets[b + 1] - offsets[b];
    const bool few = cnt < 5;                                                                     // metrics.py:83-85
    before_out[b] = cnt;
    after_out[b] = few ? 0 : num_after[b];
    tight_out[b] = few ? 0 : tight[b];
    ultra_out[b] = few ? 0 : ultra[b];
}

__device__ __forceinline__ float count_at(const void* p, int bytes, int b) {
    return bytes == 8 ? (float)reinterpret_cast<const long*>(p)[b] : (float)reinterpret_cast<const int*>(p)[b];
}

__global__ void k_pose_features(const double* __restrict__ rt, int B, const void* c0, int b0, const void* c1, int b1,
                                const void* c2, int b2, const void* c3, int b3, int width, float* __restrict__ preds,
                                float* __restrict__ inv_preds) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* m = rt + (size_t)b * 12;
    // forward pose: cast to fp32 first, then (v - mean) / std in fp32 (loftr.py:149: compute_normalized_6d(rt.float()))
    {
        const float v[9] = {(float)m[3], (float)m[7], (float)m[11], (float)m[0], (float)m[1], (float)m[2],
                            (float)m[4], (float)m[5], (float)m[6]};
#pragma unroll
        for (int j = 0; j < 9; ++j) preds[(size_t)b * width + j] = (v[j] - c_mean[j]) / c_std[j];
    }
    // inverse pose in float64 (loftr.py:147-148: linalg.inv of the 4x4), normalised in float64, then cast (:150)
    {
        const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
        const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
        const double det = a00 * c00 + a01 * c01 + a02 * c02;
        const double id = 1.0 / det;
        const double i00 = c00 * id, i01 = (a02 * a21 - a01 * a22) * id, i02 = (a01 * a12 - a02 * a11) * id;
        const double i10 = c01 * id, i11 = (a00 * a22 - a02 * a20) * id, i12 = (a02 * a10 - a00 * a12) * id;
        const double i20 = c02 * id, i21 = (a01 * a20 - a00 * a21) * id, i22 = (a00 * a11 - a01 * a10) * id;
        const double tx = m[3], ty = m[7], tz = m[11];
        const double v[9] = {-(i00 * tx + i01 * ty + i02 * tz), -(i10 * tx + i11 * ty + i12 * tz), -(i20 * tx + i21 * ty + i22 * tz),
                             i00, i01, i02, i10, i11, i12};
#pragma unroll
        for (int j = 0; j < 9; ++j) inv_preds[(size_t)b * width + j] = (float)((v[j] - (double)c_mean[j]) / (double)c_std[j]);
    }
    // correspondence counts / 500, the same columns appended to both vectors (loftr.py:158, :164-166)
    int col = 9;
    const void* cp[4] = {c0, c1, c2, c3};
    const int cb[4] = {b0, b1, b2, b3};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (cb[k]) {
            const float v = count_at(cp[k], cb[k], b) / 500.f;
            preds[(size_t)b * width + col] = v;
            inv_preds[(size_t)b * width + col] = v;
            ++col;
        }
}

}  // namespace

// prior [B][3][4] = [ rotation_6d_to_matrix(pose[3:9] std[3:9] + mean[3:9]) | pose[0:3] std[0:3] + mean[0:3] ]: the head's regressed
// pose as the next solver round's prior (loftr.py:186-192).  Gram-Schmidt as far_amd/pose6d.py does it with torch ops (F.normalize:
// x / max(|x|, 1e-12); b2 from a2 - (b1 . a2) b1; b3 = b1 x b2), in fp32.
__global__ void k_prior_from_pose(const float* __restrict__ pose, const float* __restrict__ mean, const float* __restrict__ stdv, int B,
                                  float* __restrict__ prior) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float v[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) v[e] = pose[b * 9 + e] * stdv[e] + mean[e];
    const float n1 = fmaxf(sqrtf(v[3] * v[3] + v[4] * v[4] + v[5] * v[5]), 1e-12f);
    const float b1[3] = {v[3] / n1, v[4] / n1, v[5] / n1};
    const float d = b1[0] * v[6] + b1[1] * v[7] + b1[2] * v[8];
    const float c[3] = {v[6] - d * b1[0], v[7] - d * b1[1], v[8] - d * b1[2]};
    const float n2 = fmaxf(sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), 1e-12f);
    const float b2[3] = {c[0] / n2, c[1] / n2, c[2] / n2};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    float* o = prior + b * 12;
    o[0] = b1[0]; o[1] = b1[1]; o[2] = b1[2]; o[3] = v[0];
    o[4] = b2[0]; o[5] = b2[1]; o[6] = b2[2]; o[7] = v[1];
    o[8] = b3[0]; o[9] = b3[1]; o[10] = b3[2]; o[11] = v[2];
}

extern "C" {

// pose [B][9] fp32 (normalised [t | 6D rotation]), mean / std [9] fp32 on the device -> prior [B][12] fp32 = [R | t] rows.
int far_prior_from_pose_f32(const float* pose, const float* mean, const float* stdv, int B, float* prior, hipStream_t stream) {
    far_clear_errors();
    if (B == 0) return FAR_OK;
    if (B < 0 || !pose || !mean || !stdv || !prior) return FAR_EINVAL;
    hipLaunchKernelGGL(k_prior_from_pose, dim3((B + 63) / 64), dim3(64), 0, stream, pose, mean, stdv, B, prior);
    return far_check_launch();
}

// R, E [B][9], t [B][3] float64 and status / num_after / tight / ultra [B] int32 as far_solver_f64 leaves them;
// offsets [B+1] int32 (the solver's correspondence offsets).  Outputs: rt_out [B][12] = [R | t] row-major 3x4
// (identity / zero translation where status == 0), E_out [B][9] (identity where status == 0), before_out [B] int64
// (correspondences per pair), after / tight / ultra [B] int32 (zeroed for pairs with fewer than 5 correspondences).
int far_pose_pack_f64(const double* R, const double* t, const double* E, const int* status, const int* num_after,
                      const int* tight, const int* ultra, const int* offsets, int B, double* rt_out, double* E_out,
                      long* before_out, int* after_out, int* tight_out, int* ultra_out, hipStream_t stream) {
    far_clear_errors();
    if (B == 0) return FAR_OK;
    if (B < 0 || !R || !t || !E || !status || !num_after || !tight || !ultra || !offsets || !rt_out || !E_out ||
        !before_out || !after_out || !tight_out || !ultra_out)
        return FAR_EINVAL;
    hipLaunchKernelGGL(k_pose_pack, dim3((B + 63) / 64), dim3(64), 0, stream, R, t, E, status, num_after, tight, ultra,
                       offsets, B, rt_out, E_out, before_out, after_out, tight_out, ultra_out);
    return far_check_launch();
}

// rt [B][12] float64 (3x4 row-major poses).  Up to four count vectors cnt_k [B] (elem_bytes_k = 4: int32, 8: int64,
// 0: absent; present ones first) are appended as count / 500.  preds, inv_preds: [B][9 + number of count vectors] fp32.
int far_pose_features_f32(const double* rt, int B, const void* cnt0, int elem_bytes0, const void* cnt1, int elem_bytes1,
                          const void* cnt2, int elem_bytes2, const void* cnt3, int elem_bytes3, float* preds,
                          float* inv_preds, hipStream_t stream) {
    far_clear_errors();
    if (B == 0) return FAR_OK;
    const void* cp[4] = {cnt0, cnt1, cnt2, cnt3};
    const int cb[4] = {elem_bytes0, elem_bytes1, elem_bytes2, elem_bytes3};
    int width = 9;
    for (int k = 0; k < 4; ++k) {
        if (cb[k] != 0 && cb[k] != 4 && cb[k] != 8) return FAR_EINVAL;
        if ((cb[k] != 0) != (cp[k] != nullptr)) return FAR_EINVAL;
        if (cb[k]) ++width;
    }
    if (B < 0 || !rt || !preds || !inv_preds) return FAR_EINVAL;
    hipLaunchKernelGGL(k_pose_features, dim3((B + 63) / 64), dim3(64), 0, stream, rt, B, cnt0, elem_bytes0, cnt1,
                       elem_bytes1, cnt2, elem_bytes2, cnt3, elem_bytes3, width, preds, inv_preds);
    return far_check_launch();
}

}  // extern "C"
