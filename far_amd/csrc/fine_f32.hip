// K3: fine-level window gather and sub-pixel expectation.
//
// Replaces (reference):
//   mp3d_loftr/src/loftr/loftr_module/fine_preprocess.py:40-47   F.unfold(5x5, stride 4, pad 2) of BOTH full
//        fine maps (61 MB / image) followed by the [b_ids, i_ids] gather -> here a direct gather of the
//        M x 25 x C values that are actually used;
//   mp3d_loftr/src/loftr/utils/fine_matching.py:43-54, :64-76    centre-vs-window correlation, softmax,
//        spatial expectation (kornia dsnt.spatial_expectation2d), std, mkpts1_f.
#include "common.h"

namespace {

// out[m][ww][c] = feat[b][c][y0*stride - pad + ky][x0*stride - pad + kx] (0 outside), ww = ky*W + kx,
// (y0, x0) = divmod(cell id, wc).  Arbitrary element strides so NCHW and channels_last both work; with
// channels_last (sc == 1) a wave reads 64 contiguous floats per window position.
__global__ void k_fine_gather(const float* __restrict__ feat, long sn, long sc, long sh, long sw, int C, int Hf,
                              int Wf, const int64_t* __restrict__ b_ids, const int64_t* __restrict__ cell_ids,
                              int wc, int W, int stride, int M, float* __restrict__ out) {
    const int m = blockIdx.x;
    if (m >= M) return;
    const int WW = W * W, pad = W / 2;
    const long b = b_ids[m];
    const int cell = (int)cell_ids[m];
    const int y0 = (cell / wc) * stride - pad, x0 = (cell % wc) * stride - pad;
    for (int e = threadIdx.x; e < WW * C; e += blockDim.x) {
        int ww = e / C, c = e - ww * C;
        int y = y0 + ww / W, x = x0 + ww % W;
        float v = 0.f;
        if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = feat[b * sn + (long)c * sc + (long)y * sh + (long)x * sw];
        out[((size_t)m * WW + ww) * C + c] = v;
    }
}

// Backward of k_fine_gather: dfeat[b][c][y][x] += dout[m][ww][c] (autograd of the unfold + gather of fine_preprocess.py:40-47).
// Windows overlap (5 x 5 at stride 4) and training samples cells with replacement (coarse_matching.py:216-229): fp32
// atomics, i.e. the summation ORDER is not fixed (the values are: every contribution is added exactly once).
__global__ void k_fine_scatter(const float* __restrict__ dout, long sn, long sc, long sh, long sw, int C, int Hf, int Wf,
                               const int64_t* __restrict__ b_ids, const int64_t* __restrict__ cell_ids, int wc, int W,
                               int stride, int M, float* __restrict__ dfeat) {
    const int m = blockIdx.x;
    if (m >= M) return;
    const int WW = W * W, pad = W / 2;
    const long b = b_ids[m];
    const int cell = (int)cell_ids[m];
    const int y0 = (cell / wc) * stride - pad, x0 = (cell % wc) * stride - pad;
    for (int e = threadIdx.x; e < WW * C; e += blockDim.x) {
        const int ww = e / C, c = e - ww * C;
        const int y = y0 + ww / W, x = x0 + ww % W;
        if (y >= 0 && y < Hf && x >= 0 && x < Wf)
            atomicAdd(dfeat + (b * sn + (long)c * sc + (long)y * sh + (long)x * sw), dout[((size_t)m * WW + ww) * C + c]);
    }
}

// The same backward with a FIXED summation order (the training step's other reductions are all fixed-order: K16, K6, the stem):
// pixel-centric instead of match-centric.  The host sorts the matches by (image, coarse cell) -- a stable sort of the key
// b * ncell + cell -- and passes `order` (match indices in that order) and `start` (first position of every (image, cell) group,
// Z * ncell + 1 entries).  One wave per fine-map pixel: the (at most ceil(W / stride)^2) coarse cells whose window covers it are
// visited in ascending (row, column) order, their matches in sorted order, channels across the lanes: every dfeat element is one
// thread's sequential sum.  Pixels no window covers are left as they are.
__global__ __launch_bounds__(256) void k_fine_scatter_det(const float* __restrict__ dout, long sn, long sc, long sh, long sw, int C, int Hf,
                                                          int Wf, const int64_t* __restrict__ order, const int* __restrict__ start, int wc,
                                                          int hc, int W, int stride, long npix, float* __restrict__ dfeat) {
    const long pix = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (pix >= npix) return;
    const int x = (int)(pix % Wf);
    const int y = (int)((pix / Wf) % Hf);
    const long b = pix / ((long)Wf * Hf);
    const int WW = W * W, pad = W / 2;
    // cells (cy, cx) with cy * stride - pad <= y <= cy * stride + pad
    int cy0 = (y - pad + stride - 1) / stride, cy1 = (y + pad) / stride;
    int cx0 = (x - pad + stride - 1) / stride, cx1 = (x + pad) / stride;
    cy0 = cy0 < 0 ? 0 : cy0; cx0 = cx0 < 0 ? 0 : cx0;
    cy1 = cy1 >= hc ? hc - 1 : cy1; cx1 = cx1 >= wc ? wc - 1 : cx1;
    const long ncell = (long)hc * wc;
    float* const dst = dfeat + b * sn + (long)y * sh + (long)x * sw;
    for (int c0 = lane; c0 < C; c0 += 64) {
        float acc = 0.f;
        bool any = false;
        for (int cy = cy0; cy <= cy1; ++cy)
            for (int cx = cx0; cx <= cx1; ++cx) {
                const long g = b * ncell + (long)cy * wc + cx;
                const int s0 = start[g], s1 = start[g + 1];
                const int ww = (y - (cy * stride - pad)) * W + (x - (cx * stride - pad));
                for (int s = s0; s < s1; ++s) {
                    acc += dout[((size_t)order[s] * WW + ww) * C + c0];
                    any = true;
                }
            }
        if (any) dst[(long)c0 * sc] += acc;
    }
}

// One wave per match.  feat0/feat1 [M][WW][C].  expec [M][3] = (E[x], E[y], std); mkpts1_f [M][2].
__global__ void k_fine_expect(const float* __restrict__ feat0, const float* __restrict__ feat1, int M, int W,
                              int C, const float* __restrict__ mkpts1_c, float win_scale,
                              const float* __restrict__ scale1,      // optional [Z][2] per-pair image scale
                              const int64_t* __restrict__ b_ids,     // needed only with scale1
                              float* __restrict__ expec, float* __restrict__ mkpts1_f) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= M) return;
    const int WW = W * W;
    const float* c0 = feat0 + ((size_t)wave * WW + WW / 2) * C;   // centre token (fine_matching.py:43)
    const float* f1 = feat1 + (size_t)wave * WW * C;
    const float temp = 1.0f / sqrtf((float)C);                    // :45
    // sim[r] = <centre, feat1[r]>; lanes stride the channels, then a butterfly reduction
    float sim = 0.f;  // lane r keeps sim[r] for r < WW (WW <= 64)
    for (int r = 0; r < WW; ++r) {
        float part = 0.f;
        for (int c = lane; c < C; c += 64) part += c0[c] * f1[(size_t)r * C + c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += shfl_xor_f(part, d);
        if (lane == r) sim = part;
    }
    const bool act = lane < WW;
    float x = act ? temp * sim : -FLT_MAX;
    float mx = x;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, shfl_xor_f(mx, d));
    float e = act ? expf(x - mx) : 0.f;
    float se = e;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) se += shfl_xor_f(se, d);
    float p = e / se;                                             // heatmap (:46)
    // normalised grid: linspace(-1, 1, W), x fastest (kornia create_meshgrid)
    int ky = lane / W, kx = lane - ky * W;
    float gx = act ? (W > 1 ? -1.f + 2.f * (float)kx / (float)(W - 1) : 0.f) : 0.f;
    float gy = act ? (W > 1 ? -1.f + 2.f * (float)ky / (float)(W - 1) : 0.f) : 0.f;
    float ex = p * gx, ey = p * gy, exx = p * gx * gx, eyy = p * gy * gy;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ex += shfl_xor_f(ex, d);
        ey += shfl_xor_f(ey, d);
        exx += shfl_xor_f(exx, d);
        eyy += shfl_xor_f(eyy, d);
    }
    if (lane == 0) {
        float vx = exx - ex * ex, vy = eyy - ey * ey;              // :52
        float sd = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f));  // :53
        expec[(size_t)wave * 3 + 0] = ex;
        expec[(size_t)wave * 3 + 1] = ey;
        expec[(size_t)wave * 3 + 2] = sd;
        float sx = win_scale, sy = win_scale;                      // (W // 2) * scale  (:71)
        if (scale1) { long b = b_ids[wave]; sx *= scale1[b * 2]; sy *= scale1[b * 2 + 1]; }
        mkpts1_f[(size_t)wave * 2 + 0] = mkpts1_c[(size_t)wave * 2 + 0] + ex * sx;
        mkpts1_f[(size_t)wave * 2 + 1] = mkpts1_c[(size_t)wave * 2 + 1] + ey * sy;
    }
}

}  // namespace

extern "C" {

// Gather the W x W fine windows of M matched coarse cells.  feat: fine feature map with element strides
// (sn, sc, sh, sw) and logical shape [*, C, Hf, Wf]; cell_ids index the wc-wide coarse grid; out [M][W*W][C].
int far_fine_gather_f32(const float* feat, long sn, long sc, long sh, long sw, int C, int Hf, int Wf,
                        const int64_t* b_ids, const int64_t* cell_ids, int wc, int W, int stride, int M,
                        float* out, hipStream_t stream) {
    far_clear_errors();
    if (M == 0) return FAR_OK;
    if (!feat || !b_ids || !cell_ids || !out || M < 0 || C <= 0 || W <= 0 || (W & 1) == 0 || wc <= 0) return FAR_EINVAL;
    hipLaunchKernelGGL(k_fine_gather, dim3(M), dim3(256), 0, stream, feat, sn, sc, sh, sw, C, Hf, Wf, b_ids, cell_ids,
                       wc, W, stride, M, out);
    return far_check_launch();
}

// Backward of far_fine_gather_f32: dfeat (same strides / shape as feat, zero-initialised or holding a gradient to add to)
// += the window gradients dout [M][W*W][C] at the positions the forward read.
int far_fine_scatter_f32(const float* dout, long sn, long sc, long sh, long sw, int C, int Hf, int Wf,
                         const int64_t* b_ids, const int64_t* cell_ids, int wc, int W, int stride, int M,
                         float* dfeat, hipStream_t stream) {
    far_clear_errors();
    if (M == 0) return FAR_OK;
    if (!dout || !b_ids || !cell_ids || !dfeat || M < 0 || C <= 0 || W <= 0 || (W & 1) == 0 || wc <= 0) return FAR_EINVAL;
    hipLaunchKernelGGL(k_fine_scatter, dim3(M), dim3(256), 0, stream, dout, sn, sc, sh, sw, C, Hf, Wf, b_ids, cell_ids,
                       wc, W, stride, M, dfeat);
    return far_check_launch();
}

// The same with a fixed summation order (bit-identical from run to run): `order` = the match indices sorted (stably) by
// b_ids * (hc * wc) + cell_ids, `start` = Z * hc * wc + 1 int32 offsets of the (image, cell) groups in that order (start[g + 1] -
// start[g] = matches of group g).  Z images of Hf x Wf pixels; the matches' windows are W x W at `stride` around the cells of the
// hc x wc coarse grid.
int far_fine_scatter_det_f32(const float* dout, long sn, long sc, long sh, long sw, int C, int Hf, int Wf, const int64_t* order,
                             const int* start, int Z, int hc, int wc, int W, int stride, int M, float* dfeat, hipStream_t stream) {
    far_clear_errors();
    if (M == 0 || Z == 0) return FAR_OK;
    if (!dout || !order || !start || !dfeat || M < 0 || Z < 0 || C <= 0 || W <= 0 || (W & 1) == 0 || wc <= 0 || hc <= 0 || stride <= 0 ||
        Hf <= 0 || Wf <= 0)
        return FAR_EINVAL;
    const long npix = (long)Z * Hf * Wf;
    const long blocks = (npix + 3) / 4;
    if (blocks > 0x7fffffffL) return FAR_EINVAL;
    hipLaunchKernelGGL(k_fine_scatter_det, dim3((unsigned)blocks), dim3(256), 0, stream, dout, sn, sc, sh, sw, C, Hf, Wf, order, start, wc, hc,
                       W, stride, npix, dfeat);
    return far_check_launch();
}

// Fine matching: expec_f [M][3], mkpts1_f [M][2] = mkpts1_c + E[xy] * win_scale (* scale1[b]).
int far_fine_expect_f32(const float* feat0, const float* feat1, int M, int W, int C, const float* mkpts1_c,
                        float win_scale, const float* scale1, const int64_t* b_ids, float* expec_f,
                        float* mkpts1_f, hipStream_t stream) {
    far_clear_errors();
    if (M == 0) return FAR_OK;
    if (!feat0 || !feat1 || !mkpts1_c || !expec_f || !mkpts1_f || M < 0 || W <= 0 || W * W > 64 || C <= 0)
        return FAR_EINVAL;
    if (scale1 && !b_ids) return FAR_EINVAL;
    hipLaunchKernelGGL(k_fine_expect, dim3((M + 3) / 4), dim3(256), 0, stream, feat0, feat1, M, W, C, mkpts1_c,
                       win_scale, scale1, b_ids, expec_f, mkpts1_f);
    return far_check_launch();
}

}  // extern "C"
