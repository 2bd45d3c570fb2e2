// K4: batched essential-matrix solver -- hypothesise (normalized 8-point), verify (Sampson), select,
// decompose, cheirality -- for B image pairs at once, entirely on the device, in float64.
//
// Replaces the host path of the reference (citations relative to mp3d_loftr/):
//   src/utils/metrics.py:80-174               estimate_pose (normalise by K, threshold, solver, recoverPose)
//   third_party/prior_ransac/ransac.py:340-442 RANSAC.forward: bias weights :358-371, sample :161-175,
//        estimate_model_from_minsample :251-254, remove_bad_models :303-308, get_prior_estimate :203-231,
//        verify :256-292
//   third_party/prior_ransac/cv_geometry.py:713-833  normalize_points / run_8point / normalize_transformation
//   third_party/prior_ransac/essential.py:99-139     decompose_essential_matrix
//   src/utils/cv2_fcns.py:147-319 (quoted C++)       cv::recoverPose incl. triangulatePoints
// In the reference this is 2048 cv2 calls in a Python loop plus a host<->device hop per stage; here it is six
// launches for the whole batch.  float64 because the outputs are discrete (inlier masks, argmax over
// hypotheses): with float64 the masks reproduce the oracle bit for bit, and the work is tiny
// (~0.2 GFLOP / pair) next to MI355X's 78 TFLOP/s of f64.
//
// Conventions that are this build's own (mirrored exactly by oracle/solver.py): the sampling hash, integer
// bias-weight CDF, and the sign convention of the E decomposition (right null vector has its
// largest-magnitude component positive).  See DESIGN.md "K4".
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// small dense symmetric eigen solver: cyclic Jacobi, fully unrolled, registers only
// a: upper triangle, row-major packed (idx(p,q), p <= q); v: eigenvectors in columns (v[r*N + c])
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ constexpr int tri(int p, int q) { return p * N - (p * (p - 1)) / 2 + (q - p); }

template <int N>
__device__ __forceinline__ double& sym(double (&a)[N * (N + 1) / 2], int p, int q) {
    return p <= q ? a[tri<N>(p, q)] : a[tri<N>(q, p)];
}

// eigenvector storage: registers (small N) or an LDS slab laid out [element][thread] (conflict free)
template <int N>
struct VReg {
    double v[N * N];
    __device__ __forceinline__ double get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, double x) { v[i] = x; }
};
struct VLds {
    double* base;  // &slab[threadIdx.x], element stride = blockDim.x
    int stride;
    __device__ __forceinline__ double get(int i) const { return base[i * stride]; }
    __device__ __forceinline__ void set(int i, double x) { base[i * stride] = x; }
};

// 9 x 9 eigenvectors of k_hypotheses: rows 0-7 in the LDS slab ([72][thread]: 36 KiB per 64 threads, so that FOUR workgroups fit a CU and
// the 1024 workgroups of 32 pairs x 2048 hypotheses run in one round -- [81][64] allowed three: two rounds, the second a third full),
// row 8 in registers (every index the Jacobi code passes is a compile-time constant once unrolled; the one run-time column
// lookup goes through top_dyn).
struct VHybrid {
    double* base;
    int stride;
    double t0, t1, t2, t3, t4, t5, t6, t7, t8;          // named scalars: an array member ends up in scratch
    __device__ __forceinline__ double get(int i) const {
        if (i < 72) return base[i * stride];
        switch (i) { case 72: return t0; case 73: return t1; case 74: return t2; case 75: return t3; case 76: return t4;
                     case 77: return t5; case 78: return t6; case 79: return t7; default: return t8; }
    }
    __device__ __forceinline__ void set(int i, double x) {
        if (i < 72) { base[i * stride] = x; return; }
        switch (i) { case 72: t0 = x; break; case 73: t1 = x; break; case 74: t2 = x; break; case 75: t3 = x; break; case 76: t4 = x; break;
                     case 77: t5 = x; break; case 78: t6 = x; break; case 79: t7 = x; break; default: t8 = x; break; }
    }
    __device__ __forceinline__ double top_dyn(int k) const {
        double r = t0;
        r = k == 1 ? t1 : r; r = k == 2 ? t2 : r; r = k == 3 ? t3 : r; r = k == 4 ? t4 : r;
        r = k == 5 ? t5 : r; r = k == 6 ? t6 : r; r = k == 7 ? t7 : r; r = k == 8 ? t8 : r;
        return r;
    }
};

template <int N, int P, int Q, class VS>
__device__ __forceinline__ void jacobi_rot(double (&a)[N * (N + 1) / 2], VS& v) {
    const double apq = a[tri<N>(P, Q)];
    const double app = a[tri<N>(P, P)], aqq = a[tri<N>(Q, Q)];
    double t = 0.0;
    if (apq != 0.0) {
        double theta = (aqq - app) / (2.0 * apq);
        t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
        if (theta < 0.0) t = -t;
        if (!(fabs(theta) < 1e300)) t = 0.5 / theta;  // overflow guard (theta huge -> t ~ 1/(2 theta))
        if (theta != theta) t = 0.0;
    }
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    a[tri<N>(P, P)] = app - t * apq;
    a[tri<N>(Q, Q)] = aqq + t * apq;
    a[tri<N>(P, Q)] = 0.0;
#pragma unroll
    for (int r = 0; r < N; ++r) {
        if (r != P && r != Q) {
            double& arp = sym<N>(a, r, P);
            double& arq = sym<N>(a, r, Q);
            const double x = arp, y = arq;
            arp = c * x - s * y;
            arq = s * x + c * y;
        }
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const double x = v.get(r * N + P), y = v.get(r * N + Q);
        v.set(r * N + P, c * x - s * y);
        v.set(r * N + Q, s * x + c * y);
    }
}

template <int N, int P, int Q, class VS>
struct Sweep {
    __device__ __forceinline__ static void run(double (&a)[N * (N + 1) / 2], VS& v) {
        jacobi_rot<N, P, Q, VS>(a, v);
        if constexpr (Q + 1 < N) Sweep<N, P, Q + 1, VS>::run(a, v);
        else if constexpr (P + 2 < N) Sweep<N, P + 1, P + 2, VS>::run(a, v);
    }
};

template <int N, class VS>
__device__ __forceinline__ void jacobi_eig(double (&a)[N * (N + 1) / 2], VS& v, int max_sweeps) {
#pragma unroll
    for (int i = 0; i < N * N; ++i) v.set(i, (i / N == i % N) ? 1.0 : 0.0);
    for (int sw = 0; sw < max_sweeps; ++sw) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int p = 0; p < N; ++p) {
            dg += a[tri<N>(p, p)] * a[tri<N>(p, p)];
#pragma unroll
            for (int q = p + 1; q < N; ++q) off += a[tri<N>(p, q)] * a[tri<N>(p, q)];
        }
        if (!(off > 1e-50 * dg)) break;
        Sweep<N, 0, 1, VS>::run(a, v);
    }
}

// 3x3: eigen decomposition of M^T M sorted by descending eigenvalue. V columns = right singular vectors.
__device__ __forceinline__ void right_singular_3x3(const double (&M)[9], double (&V)[9], double (&lam)[3]) {
    double a[6];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = p; q < 3; ++q) a[tri<3>(p, q)] = M[p] * M[q] + M[3 + p] * M[3 + q] + M[6 + p] * M[6 + q];
    VReg<3> vv;
    jacobi_eig<3>(a, vv, 30);
    const double (&v)[9] = vv.v;
    double l0 = a[tri<3>(0, 0)], l1 = a[tri<3>(1, 1)], l2 = a[tri<3>(2, 2)];
    // sorting network on (lambda, column)
    int i0 = 0, i1 = 1, i2 = 2;
    if (l0 < l1) { double t = l0; l0 = l1; l1 = t; int k = i0; i0 = i1; i1 = k; }
    if (l1 < l2) { double t = l1; l1 = l2; l2 = t; int k = i1; i1 = i2; i2 = k; }
    if (l0 < l1) { double t = l0; l0 = l1; l1 = t; int k = i0; i0 = i1; i1 = k; }
    lam[0] = l0; lam[1] = l1; lam[2] = l2;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double c0 = v[r * 3 + 0], c1 = v[r * 3 + 1], c2 = v[r * 3 + 2];
        V[r * 3 + 0] = i0 == 0 ? c0 : (i0 == 1 ? c1 : c2);
        V[r * 3 + 1] = i1 == 0 ? c0 : (i1 == 1 ? c1 : c2);
        V[r * 3 + 2] = i2 == 0 ? c0 : (i2 == 1 ? c1 : c2);
    }
}

// E -> R1, R2, t with this build's sign convention (see header comment / oracle decompose_essential).
__device__ __forceinline__ void decompose_E(const double (&E)[9], double (&R1)[9], double (&R2)[9], double (&t)[3]) {
    double V[9], lam[3];
    right_singular_3x3(E, V, lam);
    // v3: largest-|component| positive (ties -> lowest index)
    double v30 = V[2], v31 = V[5], v32 = V[8];
    int k = 0; double big = fabs(v30);
    if (fabs(v31) > big) { big = fabs(v31); k = 1; }
    if (fabs(v32) > big) { big = fabs(v32); k = 2; }
    double vk = k == 0 ? v30 : (k == 1 ? v31 : v32);
    if (vk < 0.0) { V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8]; }
    // right-handed: det([v1 v2 v3]) > 0, else flip v1
    double det = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    if (det < 0.0) { V[0] = -V[0]; V[3] = -V[3]; V[6] = -V[6]; }
    double u1[3], u2[3], u3[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        u1[r] = E[r * 3 + 0] * V[0] + E[r * 3 + 1] * V[3] + E[r * 3 + 2] * V[6];
        u2[r] = E[r * 3 + 0] * V[1] + E[r * 3 + 1] * V[4] + E[r * 3 + 2] * V[7];
    }
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) { u1[r] /= n1; u2[r] /= n2; }
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
    u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
    u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    // U W V^T with W = [[0,-1,0],[1,0,0],[0,0,1]]:  U W = [u2, -u1, u3];  U W^T = [-u2, u1, u3]
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double a = u2[r] * V[c * 3 + 0] - u1[r] * V[c * 3 + 1];
            double b = u3[r] * V[c * 3 + 2];
            R1[r * 3 + c] = a + b;
            R2[r * 3 + c] = -a + b;
        }
    t[0] = u3[0]; t[1] = u3[1]; t[2] = u3[2];
}

// ---------------------------------------------------------------------------------------------
// sampling hash (oracle/solver.py: mix32 / hash_u32)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t hash_u32(uint32_t seed, uint32_t b, uint32_t h, uint32_t s) {
    uint32_t key = mix32(seed ^ b);
    key = mix32(key + h * 0x9E3779B1u);
    return mix32(key + s * 0x85EBCA77u);
}

struct PairPrior {       // per pair, prior mode
    double E[9];         // [t]x R of the normalised prior
    double RT[12];       // R | t/|t|
};

// ---------------------------------------------------------------------------------------------
// 1. prepare: normalised coordinates (float64), their float32 roundings, bias weights
// ---------------------------------------------------------------------------------------------
__global__ void k_prior_setup(const float* __restrict__ priorRT, int B, PairPrior* __restrict__ pp,
                              const float* __restrict__ pcl, int P, double* __restrict__ tgt) {
    const int b = blockIdx.x;
    __shared__ PairPrior sp;
    if (threadIdx.x == 0) {
        double R[9], t[3];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = (double)priorRT[b * 12 + r * 4 + c];
            t[r] = (double)priorRT[b * 12 + r * 4 + 3];
        }
        double n = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        t[0] /= n; t[1] /= n; t[2] /= n;                                   // ransac.py:183
        const double Tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                sp.E[r * 3 + c] = Tx[r * 3] * R[c] + Tx[r * 3 + 1] * R[3 + c] + Tx[r * 3 + 2] * R[6 + c];
                sp.RT[r * 4 + c] = R[r * 3 + c];
            }
        sp.RT[3] = t[0]; sp.RT[7] = t[1]; sp.RT[11] = t[2];
        pp[b] = sp;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        double x = pcl[p * 3], y = pcl[p * 3 + 1], z = pcl[p * 3 + 2];
        for (int r = 0; r < 3; ++r)
            tgt[((size_t)b * P + p) * 3 + r] = sp.RT[r * 4] * x + sp.RT[r * 4 + 1] * y + sp.RT[r * 4 + 2] * z + sp.RT[r * 4 + 3];
    }
}

__device__ __forceinline__ double sym_epi(const double (&F)[9], double x1, double y1, double x2, double y2) {
    double l1x = F[0] * x1 + F[1] * y1 + F[2], l1y = F[3] * x1 + F[4] * y1 + F[5], l1z = F[6] * x1 + F[7] * y1 + F[8];
    double l2x = F[0] * x2 + F[3] * y2 + F[6], l2y = F[1] * x2 + F[4] * y2 + F[7];
    double num = x2 * l1x + y2 * l1y + l1z;
    num *= num;
    return num * (1.0 / (l1x * l1x + l1y * l1y) + 1.0 / (l2x * l2x + l2y * l2y));
}

__device__ __forceinline__ double sampson(const double (&F)[9], double x1, double y1, double x2, double y2) {
    double l1x = F[0] * x1 + F[1] * y1 + F[2], l1y = F[3] * x1 + F[4] * y1 + F[5], l1z = F[6] * x1 + F[7] * y1 + F[8];
    double l2x = F[0] * x2 + F[3] * y2 + F[6], l2y = F[1] * x2 + F[4] * y2 + F[7];
    double num = x2 * l1x + y2 * l1y + l1z;
    num *= num;
    return num / (l1x * l1x + l1y * l1y + l2x * l2x + l2y * l2y);
}

__global__ void k_prepare(const float* __restrict__ kpts0, const float* __restrict__ kpts1,
                          const int* __restrict__ offsets, const double* __restrict__ K0,
                          const double* __restrict__ K1, const PairPrior* __restrict__ pp,
                          double4* __restrict__ kn, double4* __restrict__ kp, uint32_t* __restrict__ wq) {
    const int b = blockIdx.y;
    const int o = offsets[b], M = offsets[b + 1] - o;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double* k0 = K0 + b * 9;
    const double* k1 = K1 + b * 9;
    double x0 = ((double)kpts0[2 * (o + i)] - k0[2]) / k0[0];          // metrics.py:88
    double y0 = ((double)kpts0[2 * (o + i) + 1] - k0[5]) / k0[4];
    double x1 = ((double)kpts1[2 * (o + i)] - k1[2]) / k1[0];          // :89
    double y1 = ((double)kpts1[2 * (o + i) + 1] - k1[5]) / k1[4];
    kn[o + i] = make_double4(x0, y0, x1, y1);
    double4 r = make_double4((double)(float)x0, (double)(float)y0, (double)(float)x1, (double)(float)y1);
    kp[o + i] = r;                                                       // torch.FloatTensor(kpts_norm) :124
    if (pp) {
        double E[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) E[e] = pp[b].E[e];
        double d = sym_epi(E, r.x, r.y, r.z, r.w);                       // ransac.py:364
        double w = exp(-d / 0.1);                                        // :366, bias_sigma_sq = 0.1
        if (!(w == w) || !(fabs(w) < 1e300)) w = 0.0;
        double q = floor((w + 1e-4) * 65536.0);                          // integer weights (oracle quantize_weights)
        wq[o + i] = q < 1.0 ? 1u : (uint32_t)q;
    }
}

// inclusive scan of integer weights, one block per pair (M <= a few thousand)
__global__ void k_cdf(const uint32_t* __restrict__ wq, const int* __restrict__ offsets, uint32_t* __restrict__ cdf) {
    __shared__ uint32_t part[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const int o = offsets[b], M = offsets[b + 1] - o;
    const int per = (M + 255) / 256;
    const int s = t * per, e = min(M, s + per);
    uint32_t acc = 0;
    for (int i = s; i < e; ++i) acc += wq[o + i];
    part[t] = acc;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; ++i) { uint32_t x = part[i]; part[i] = run; run += x; }
    }
    __syncthreads();
    acc = part[t];
    for (int i = s; i < e; ++i) { acc += wq[o + i]; cdf[o + i] = acc; }
}

// ---------------------------------------------------------------------------------------------
// 2. hypotheses: one thread per (pair, hypothesis)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sample8(uint32_t seed, int b, int h, int M, const uint32_t* __restrict__ cdf,
                                        int (&idx)[8]) {
    if (cdf) {
        const uint32_t total = cdf[M - 1];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint32_t r = hash_u32(seed, b, h, s) % total;
            int lo = 0, hi = M - 1;                 // first i with cdf[i] > r
            while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] > r) hi = mid; else lo = mid + 1; }
            idx[s] = lo;
        }
    } else {
        int srt[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int r = (int)(hash_u32(seed, b, h, s) % (uint32_t)(M - s));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < s && srt[j] <= r) ++r;
            idx[s] = r;
            // insert r into the sorted prefix srt[0..s)
            int v = r;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < s && srt[j] > v) { int tmp = srt[j]; srt[j] = v; v = tmp; }
            srt[s] = v;
        }
    }
}

// ransac.py:203-231: mean |[R|t] x - target| over the point cloud, min over R1 / R2; :397 -err^2 / lambda
__device__ __forceinline__ double prior_score_of(const double (&Fe)[9], int b, const float* __restrict__ pcl,
                                                 const double* __restrict__ tgt, int P, double lambda) {
    double R1[9], R2[9], t[3];
    decompose_E(Fe, R1, R2, t);
    double e1 = 0.0, e2 = 0.0;
    const double* tg = tgt + (size_t)b * P * 3;
    for (int p = 0; p < P; ++p) {
        const double x = pcl[p * 3], y = pcl[p * 3 + 1], z = pcl[p * 3 + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double g = tg[p * 3 + r];
            e1 += fabs(R1[r * 3] * x + R1[r * 3 + 1] * y + R1[r * 3 + 2] * z + t[r] - g);
            e2 += fabs(R2[r * 3] * x + R2[r * 3 + 1] * y + R2[r * 3 + 2] * z + t[r] - g);
        }
    }
    const double err = fmin(e1, e2) / (3.0 * P);
    return -(err * err) / lambda;
}

// The normalized 8-point from the 9 x 9 normal matrix on (cv_geometry.py:814-833): null vector (Jacobi), rank-2 projection,
// de-normalisation with the two Hartley transforms T = [[s, 0, tx], [0, s, ty], [0, 0, 1]], normalize_transformation.
// Shared by k_hypotheses (8 sampled correspondences) and k_eightpoint (run_8point's N weighted correspondences).
__device__ __forceinline__ void eightpoint_from_normal(double (&a)[45], VHybrid& v, double s1, double t1x, double t1y,
                                                       double s2, double t2x, double t2y, double (&Fe)[9]) {
    jacobi_eig<9>(a, v, 20);
    // eigenvector of the smallest eigenvalue (cv_geometry.py:820-821)
    int km = 0; double lm = a[tri<9>(0, 0)];
#pragma unroll
    for (int p = 1; p < 9; ++p) { double l = a[tri<9>(p, p)]; if (l < lm) { lm = l; km = p; } }
    double Fm[9];
#pragma unroll
    for (int r = 0; r < 8; ++r) Fm[r] = v.base[(r * 9 + km) * v.stride];
    Fm[8] = v.top_dyn(km);
    // rank-2 projection: F - (F v3) v3^T  == U diag(s1, s2, 0) V^T  (cv_geometry.py:824-827)
    double V3[9], lam[3];
    right_singular_3x3(Fm, V3, lam);
    {
        const double n0 = V3[2], n1 = V3[5], n2 = V3[8];
        double Fv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) Fv[r] = Fm[r * 3] * n0 + Fm[r * 3 + 1] * n1 + Fm[r * 3 + 2] * n2;
#pragma unroll
        for (int r = 0; r < 3; ++r) { Fm[r * 3] -= Fv[r] * n0; Fm[r * 3 + 1] -= Fv[r] * n1; Fm[r * 3 + 2] -= Fv[r] * n2; }
    }
    // F_est = T2^T Fp T1 (cv_geometry.py:828), T = [[s,0,tx],[0,s,ty],[0,0,1]]
    double G[9];  // Fp T1
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        G[r * 3 + 0] = Fm[r * 3 + 0] * s1;
        G[r * 3 + 1] = Fm[r * 3 + 1] * s1;
        G[r * 3 + 2] = Fm[r * 3 + 0] * t1x + Fm[r * 3 + 1] * t1y + Fm[r * 3 + 2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Fe[0 + c] = s2 * G[0 + c];
        Fe[3 + c] = s2 * G[3 + c];
        Fe[6 + c] = t2x * G[0 + c] + t2y * G[3 + c] + G[6 + c];
    }
    // normalize_transformation (cv_geometry.py:753-769)
    const double nv = Fe[8];
    if (fabs(nv) > 1e-8) {
#pragma unroll
        for (int e = 0; e < 9; ++e) Fe[e] = Fe[e] / (nv + 1e-8);
    }
}

__global__ __launch_bounds__(64, 1) void k_hypotheses(
    const double4* __restrict__ kp, const int* __restrict__ offsets, const uint32_t* __restrict__ cdf_all,
    const int* __restrict__ samples_in, int H, int minimal, uint32_t seed, const PairPrior* __restrict__ pp,
    const float* __restrict__ pcl, const double* __restrict__ tgt, int P, double lambda,
    double* __restrict__ F_all, double* __restrict__ pscore, int* __restrict__ samples_out) {
    const int b = blockIdx.y, h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    const int o = offsets[b], M = offsets[b + 1] - o;
    const size_t hid = (size_t)b * H + h;
    if (minimal == 5 || (M >= 5 && M < 8)) return;          // this pair's hypotheses come from k_hypotheses5
    if (M < 8) { pscore[hid] = -INFINITY; for (int e = 0; e < 9; ++e) F_all[hid * 9 + e] = 0.0; return; }
    int idx[8];
    if (samples_in) {
#pragma unroll
        for (int s = 0; s < 8; ++s) idx[s] = samples_in[hid * 8 + s];
    } else {
        sample8(seed, b, h, M, cdf_all ? cdf_all + o : nullptr, idx);
    }
    if (samples_out) {
#pragma unroll
        for (int s = 0; s < 8; ++s) samples_out[hid * 8 + s] = idx[s];
    }
    double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { double4 p = kp[o + idx[s]]; x1[s] = p.x; y1[s] = p.y; x2[s] = p.z; y2[s] = p.w; }
    // Hartley normalisation (cv_geometry.py:713-750)
    double m1x = 0, m1y = 0, m2x = 0, m2y = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) { m1x += x1[s]; m1y += y1[s]; m2x += x2[s]; m2y += y2[s]; }
    m1x /= 8.0; m1y /= 8.0; m2x /= 8.0; m2y /= 8.0;
    double d1 = 0, d2 = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        d1 += sqrt((x1[s] - m1x) * (x1[s] - m1x) + (y1[s] - m1y) * (y1[s] - m1y));
        d2 += sqrt((x2[s] - m2x) * (x2[s] - m2x) + (y2[s] - m2y) * (y2[s] - m2y));
    }
    const double s1 = sqrt(2.0) / (d1 / 8.0 + 1e-8), s2 = sqrt(2.0) / (d2 / 8.0 + 1e-8);
    const double t1x = -s1 * m1x, t1y = -s1 * m1y, t2x = -s2 * m2x, t2y = -s2 * m2y;
    // A = X^T X, X rows [x2x1, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1]  (cv_geometry.py:810-814)
    double a[45];
#pragma unroll
    for (int e = 0; e < 45; ++e) a[e] = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const double ax = x1[s] * s1 + t1x, ay = y1[s] * s1 + t1y, bx = x2[s] * s2 + t2x, by = y2[s] * s2 + t2y;
        const double row[9] = {bx * ax, bx * ay, bx, by * ax, by * ay, by, ax, ay, 1.0};
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int q = p; q < 9; ++q) a[tri<9>(p, q)] += row[p] * row[q];
    }
    extern __shared__ __attribute__((aligned(16))) double vslab[];   // [72][blockDim.x]
    VHybrid v;
    v.base = vslab + threadIdx.x; v.stride = (int)blockDim.x;
    double Fe[9];
    eightpoint_from_normal(a, v, s1, t1x, t1y, s2, t2x, t2y, Fe);
    bool finite = true;
#pragma unroll
    for (int e = 0; e < 9; ++e) finite = finite && (fabs(Fe[e]) < 1e300);
    const double dmin = fmin(fabs(Fe[0]), fmin(fabs(Fe[4]), fabs(Fe[8])));
    bool distinct = true;   // repeated correspondence -> rank-deficient system -> rejected (oracle/solver.py)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = i + 1; j < 8; ++j) distinct = distinct && (idx[i] != idx[j]);
    const bool valid = finite && distinct && (dmin > 1e-4);           // ransac.py:306-307
#pragma unroll
    for (int e = 0; e < 9; ++e) F_all[hid * 9 + e] = Fe[e];
    double ps = 0.0;
    if (valid && pp) ps = prior_score_of(Fe, b, pcl, tgt, P, lambda);
    pscore[hid] = valid ? ps : -INFINITY;
}


// run_8point (cv_geometry.py:772-833) as a function of its own: B problems of N >= 8 weighted correspondences each, one thread
// per problem (the function-level API of far_amd/ransac.py; K4's hypothesis stage above is the same arithmetic on 8 sampled points).
// p1, p2 [B][N][2], w [B][N] or null (all ones), F_out [B][9]; float64.
__global__ __launch_bounds__(64, 1) void k_eightpoint(const double* __restrict__ p1, const double* __restrict__ p2,
                                                      const double* __restrict__ w, int B, int N, double* __restrict__ F_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* a1 = p1 + (size_t)b * N * 2;
    const double* a2 = p2 + (size_t)b * N * 2;
    double m1x = 0, m1y = 0, m2x = 0, m2y = 0;
    for (int i = 0; i < N; ++i) { m1x += a1[2 * i]; m1y += a1[2 * i + 1]; m2x += a2[2 * i]; m2y += a2[2 * i + 1]; }
    m1x /= N; m1y /= N; m2x /= N; m2y /= N;
    double d1 = 0, d2 = 0;
    for (int i = 0; i < N; ++i) {
        d1 += sqrt((a1[2 * i] - m1x) * (a1[2 * i] - m1x) + (a1[2 * i + 1] - m1y) * (a1[2 * i + 1] - m1y));
        d2 += sqrt((a2[2 * i] - m2x) * (a2[2 * i] - m2x) + (a2[2 * i + 1] - m2y) * (a2[2 * i + 1] - m2y));
    }
    const double s1 = sqrt(2.0) / (d1 / N + 1e-8), s2 = sqrt(2.0) / (d2 / N + 1e-8);
    const double t1x = -s1 * m1x, t1y = -s1 * m1y, t2x = -s2 * m2x, t2y = -s2 * m2y;
    double a[45];
#pragma unroll
    for (int e = 0; e < 45; ++e) a[e] = 0.0;
    for (int i = 0; i < N; ++i) {
        const double ax = a1[2 * i] * s1 + t1x, ay = a1[2 * i + 1] * s1 + t1y, bx = a2[2 * i] * s2 + t2x, by = a2[2 * i + 1] * s2 + t2y;
        const double wi = w ? w[(size_t)b * N + i] : 1.0;
        const double row[9] = {bx * ax, bx * ay, bx, by * ax, by * ay, by, ax, ay, 1.0};
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int q = p; q < 9; ++q) a[tri<9>(p, q)] += (row[p] * wi) * row[q];         // X^T diag(w) X (:814-817)
    }
    extern __shared__ __attribute__((aligned(16))) double vslab[];   // [72][blockDim.x]
    VHybrid v;
    v.base = vslab + threadIdx.x; v.stride = (int)blockDim.x;
    double Fe[9];
    eightpoint_from_normal(a, v, s1, t1x, t1y, s2, t2x, t2y, Fe);
#pragma unroll
    for (int e = 0; e < 9; ++e) F_out[(size_t)b * 9 + e] = Fe[e];
}

__global__ void k_identity9(double* __restrict__ K, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
#pragma unroll
    for (int e = 0; e < 9; ++e) K[(size_t)b * 9 + e] = (e % 4 == 0) ? 1.0 : 0.0;
}

// decompose_essential_matrix (essential.py:99-139) for n matrices: R1 = U W V^T, R2 = U W^T V^T, t = u3, with this build's sign
// convention (decompose_E above; the reference's follows LAPACK's -- the set {R1, R2} x {t, -t} is the same).
__global__ void k_decompose(const double* __restrict__ E, long n, double* __restrict__ R1o, double* __restrict__ R2o, double* __restrict__ to) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double Em[9], R1[9], R2[9], t[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) Em[e] = E[i * 9 + e];
    decompose_E(Em, R1, R2, t);
#pragma unroll
    for (int e = 0; e < 9; ++e) { R1o[i * 9 + e] = R1[e]; R2o[i * 9 + e] = R2[e]; }
    to[i * 3] = t[0]; to[i * 3 + 1] = t[1]; to[i * 3 + 2] = t[2];
}

#include "solver5_f64.inc"

// ---------------------------------------------------------------------------------------------
// 3. verify: count Sampson inliers of every hypothesis (ransac.py:256-292)
// ---------------------------------------------------------------------------------------------
constexpr int SC_PTS = 512;

// One workgroup = 64 hypotheses x 4 parts: the four threads of a hypothesis count the inliers of every fourth staged correspondence
// and the (integer, order-independent) counts are added through LDS -- a quarter of the serial latency of one thread per
// hypothesis (215-285 us per round at ~1 900 correspondences, whatever the batch).
__global__ __launch_bounds__(256) void k_score(const double4* __restrict__ kp, const int* __restrict__ offsets,
                        const double* __restrict__ F_all, const double* __restrict__ pscore,
                        const double* __restrict__ inl_th, int H, int* __restrict__ count_all,
                        double* __restrict__ score_all) {
    __shared__ double4 pts[SC_PTS];
    __shared__ int part_cnt[3][64];
    const int hl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int b = blockIdx.y, h = blockIdx.x * 64 + hl;
    const int o = offsets[b], M = offsets[b + 1] - o;
    const size_t hid = (size_t)b * H + (h < H ? h : H - 1);
    double F[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) F[e] = F_all[hid * 9 + e];
    const double thr = inl_th[b];
    int cnt = 0;
    for (int p0 = 0; p0 < M; p0 += SC_PTS) {
        const int np = min(SC_PTS, M - p0);
        __syncthreads();
        for (int i = threadIdx.x; i < np; i += blockDim.x) pts[i] = kp[o + p0 + i];
        __syncthreads();
        for (int i = part; i < np; i += 4) {
            const double4 p = pts[i];
            cnt += (sampson(F, p.x, p.y, p.z, p.w) <= thr) ? 1 : 0;
        }
    }
    if (part) part_cnt[part - 1][hl] = cnt;
    __syncthreads();
    if (part == 0 && h < H) {
        cnt += part_cnt[0][hl] + part_cnt[1][hl] + part_cnt[2][hl];
        const double ps = pscore[hid];
        count_all[hid] = cnt;
        score_all[hid] = (ps == -INFINITY) ? -INFINITY : (double)cnt + ps;
    }
}

// ---------------------------------------------------------------------------------------------
// 4. select the best hypothesis (first maximum), its inlier masks at thr, thr/10, thr/100
// ---------------------------------------------------------------------------------------------
__global__ void k_select(const double4* __restrict__ kp, const int* __restrict__ offsets,
                         const double* __restrict__ F_all, const double* __restrict__ score_all,
                         const double* __restrict__ inl_th, int H, int many_thr, int minimal,
                         int* __restrict__ best_out, double* __restrict__ E_out, uint8_t* __restrict__ mask,
                         int* __restrict__ n_inl, int* __restrict__ n_tight, int* __restrict__ n_ultra,
                         int* __restrict__ status, int* __restrict__ good_g, int mask_bits) {
    __shared__ double sv[256];
    __shared__ int si[256];
    __shared__ int cnt[3];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < 4) good_g[b * 4 + t] = 0;                      // the cheirality counters of k_recover_bits (next launch)
    const int o = offsets[b], M = offsets[b + 1] - o;
    double bv = -INFINITY; int bi = 0x7fffffff;
    for (int h = t; h < H; h += 256) {
        double s = score_all[(size_t)b * H + h];
        if (s > bv) { bv = s; bi = h; }          // strided: smaller h seen first per thread
    }
    sv[t] = bv; si[t] = bi;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (t < d) {
            double ov = sv[t + d]; int oi = si[t + d];
            if (ov > sv[t] || (ov == sv[t] && oi < si[t])) { sv[t] = ov; si[t] = oi; }
        }
        __syncthreads();
    }
    const double best_score = sv[0];
    const int best = si[0];
    // ransac.py:353, :409: the best score must exceed the minimal sample size of the solver that produced the models
    const bool five = minimal == 5 || M < 8;
    const bool ok = M >= (five ? 5 : 8) && best != 0x7fffffff && best_score > (five ? 5.0 : 8.0);
    if (t < 3) cnt[t] = 0;
    __syncthreads();
    double F[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) F[e] = ok ? F_all[((size_t)b * H + best) * 9 + e] : 0.0;
    const double thr = inl_th[b];
    int c0 = 0, c1 = 0, c2 = 0;
    for (int i = t; i < M; i += 256) {
        uint8_t m = 0;
        if (ok) {
            const double4 p = kp[o + i];
            const double e = sampson(F, p.x, p.y, p.z, p.w);
            m = e <= thr;
            const bool tight = e <= thr / 10.0, ultra = e <= thr / 100.0;  // ransac.py:284-285
            c0 += m;
            c1 += tight;
            c2 += ultra;
            if (mask_bits) m |= (tight ? 2 : 0) | (ultra ? 4 : 0);         // far_ransac_f64: the three masks RANSAC.forward returns
        }
        mask[o + i] = m;
    }
    atomicAdd(&cnt[0], c0); atomicAdd(&cnt[1], c1); atomicAdd(&cnt[2], c2);
    __syncthreads();
    if (t == 0) {
        best_out[b] = ok ? best : -1;
        n_inl[b] = cnt[0];
        n_tight[b] = many_thr ? cnt[1] : 0;
        n_ultra[b] = many_thr ? cnt[2] : 0;
        status[b] = ok ? 1 : 0;
#pragma unroll
        for (int e = 0; e < 9; ++e) E_out[b * 9 + e] = F[e];
    }
}

// ---------------------------------------------------------------------------------------------
// 5. cheirality: cv::recoverPose (cv2_fcns.py:147-319) with K = I, distanceThresh = dist
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t cheirality_bits(const double (&R1)[9], const double (&R2)[9], const double (&t)[3],
                                                   double x0, double y0, double x1, double y1, double dist) {
    uint8_t bits = 0;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        const bool second = (c & 1) != 0;
        const double sg = (c & 2) ? -1.0 : 1.0;
        double P[12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            P[r * 4] = second ? R2[r * 3] : R1[r * 3];
            P[r * 4 + 1] = second ? R2[r * 3 + 1] : R1[r * 3 + 1];
            P[r * 4 + 2] = second ? R2[r * 3 + 2] : R1[r * 3 + 2];
            P[r * 4 + 3] = sg * t[r];
        }
        // DLT rows (cv::triangulatePoints): x*P[2]-P[0], y*P[2]-P[1] for both views; P0 = [I|0]
        double A[16];
        A[0] = -1.0; A[1] = 0.0; A[2] = x0; A[3] = 0.0;
        A[4] = 0.0; A[5] = -1.0; A[6] = y0; A[7] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { A[8 + k] = x1 * P[8 + k] - P[k]; A[12 + k] = y1 * P[8 + k] - P[4 + k]; }
        double g[10];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p; q < 4; ++q)
                g[tri<4>(p, q)] = A[p] * A[q] + A[4 + p] * A[4 + q] + A[8 + p] * A[8 + q] + A[12 + p] * A[12 + q];
        VReg<4> vv;
        jacobi_eig<4>(g, vv, 30);
        const double (&v)[16] = vv.v;
        int km = 0; double lm = g[tri<4>(0, 0)];
#pragma unroll
        for (int p = 1; p < 4; ++p) { double l = g[tri<4>(p, p)]; if (l < lm) { lm = l; km = p; } }
        double Q[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double acc = v[r * 4];
#pragma unroll
            for (int cc = 1; cc < 4; ++cc) acc = (km == cc) ? v[r * 4 + cc] : acc;
            Q[r] = acc;
        }
        bool m = (Q[2] * Q[3]) > 0.0;
        const double X = Q[0] / Q[3], Y = Q[1] / Q[3], Z = Q[2] / Q[3];
        m = m && (Z < dist);
        const double z2 = P[8] * X + P[9] * Y + P[10] * Z + P[11];
        m = m && (z2 > 0.0) && (z2 < dist);
        bits |= (m ? 1 : 0) << c;
    }
    return bits;
}

// recoverPose in two launches.  k_recover_bits: the four candidate poses' cheirality tests -- a 4 x 4 eigen decomposition per
// (correspondence, candidate), cv::triangulatePoints' DLT -- one correspondence per thread, ceil(M / 256) workgroups per pair (as one
// workgroup per pair, 32 pairs kept 32 CUs busy for 0.37 ms with ~30 serial decompositions per thread); the per-candidate counts
// are integer atomics (order-free).  k_recover_final: the candidate with the most points in front, the mask rewritten in place, outputs.
__global__ __launch_bounds__(256) void k_recover_bits(const double4* __restrict__ kn, const int* __restrict__ offsets,
                          const double* __restrict__ E_in, const int* __restrict__ status_in, double dist,
                          const uint8_t* __restrict__ mask, uint8_t* __restrict__ bits_ws, int* __restrict__ good_g,
                          double* __restrict__ rt_ws) {
    __shared__ int good[4];
    const int b = blockIdx.y, tdx = threadIdx.x;
    const int o = offsets[b], M = offsets[b + 1] - o;
    if ((int)blockIdx.x * 256 >= M && blockIdx.x != 0) return;            // (workgroup 0 always runs: it hands on R1, R2, t)
    if (tdx < 4) good[tdx] = 0;
    __syncthreads();
    const bool ok = status_in[b] != 0;
    double E[9], R1[9], R2[9], t[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) E[e] = E_in[b * 9 + e];
    if (ok) decompose_E(E, R1, R2, t);
    else {
#pragma unroll
        for (int e = 0; e < 9; ++e) { R1[e] = (e % 4 == 0); R2[e] = R1[e]; }
        t[0] = t[1] = t[2] = 0.0;
    }
    if (blockIdx.x == 0 && tdx == 0) {
        double* w = rt_ws + (size_t)b * 21;
#pragma unroll
        for (int e = 0; e < 9; ++e) { w[e] = R1[e]; w[9 + e] = R2[e]; }
        w[18] = t[0]; w[19] = t[1]; w[20] = t[2];
    }
    const int i = blockIdx.x * 256 + tdx;
    if (ok && i < M) {
        const double4 p = kn[o + i];
        uint8_t bits = cheirality_bits(R1, R2, t, p.x, p.y, p.z, p.w, dist);
        if (!mask[o + i]) bits = 0;                                       // AND with the RANSAC mask
        bits_ws[o + i] = bits;
        if (bits & 1) atomicAdd(&good[0], 1);
        if (bits & 2) atomicAdd(&good[1], 1);
        if (bits & 4) atomicAdd(&good[2], 1);
        if (bits & 8) atomicAdd(&good[3], 1);
    }
    __syncthreads();
    if (tdx < 4 && good[tdx]) atomicAdd(&good_g[b * 4 + tdx], good[tdx]);
}

__global__ __launch_bounds__(256) void k_recover_final(const int* __restrict__ offsets, const int* __restrict__ status_in,
                          const int* __restrict__ good_g, const double* __restrict__ rt_ws, uint8_t* __restrict__ mask,
                          const uint8_t* __restrict__ bits_ws, double* __restrict__ R_out, double* __restrict__ t_out,
                          int* __restrict__ n_out, int* __restrict__ num_after, int* __restrict__ status_out) {
    __shared__ int after;
    const int b = blockIdx.x, tdx = threadIdx.x;
    const int o = offsets[b], M = offsets[b + 1] - o;
    if (tdx == 0) after = 0;
    __syncthreads();
    const bool ok = status_in[b] != 0;
    const int a = good_g[b * 4], bb = good_g[b * 4 + 1], c = good_g[b * 4 + 2], d = good_g[b * 4 + 3];
    int ch;
    if (a >= bb && a >= c && a >= d) ch = 0;
    else if (bb >= a && bb >= c && bb >= d) ch = 1;
    else if (c >= a && c >= bb && c >= d) ch = 2;
    else ch = 3;
    int aft = 0;
    for (int i = tdx; i < M; i += blockDim.x) {
        uint8_t m = ok ? ((bits_ws[o + i] >> ch) & 1) : 0;
        mask[o + i] = m;                                                  // recoverPose rewrites the mask in place
        aft += m;
    }
    atomicAdd(&after, aft);
    __syncthreads();
    if (tdx == 0) {
        const double* w = rt_ws + (size_t)b * 21;
        const double* R = (ch & 1) ? w + 9 : w;
        const double sg = (ch & 2) ? -1.0 : 1.0;
#pragma unroll
        for (int e = 0; e < 9; ++e) R_out[b * 9 + e] = R[e];
        t_out[b * 3] = sg * w[18]; t_out[b * 3 + 1] = sg * w[19]; t_out[b * 3 + 2] = sg * w[20];
        const int n = ok ? (ch == 0 ? a : ch == 1 ? bb : ch == 2 ? c : d) : 0;
        n_out[b] = n;
        num_after[b] = after;
        status_out[b] = (ok && n > 0) ? 1 : 0;                            // metrics.py:166 (n > best_num_inliers = 0)
    }
}

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct SolverWs {
    double4* kn; double4* kp; uint32_t* wq; uint32_t* cdf; PairPrior* pp; double* tgt;
    double* F_all; double* pscore; int* count_all; double* score_all; uint8_t* bits; int* status_sel; int* n_inl;
    int* good; double* rt;
    size_t bytes;
};

SolverWs carve(void* ws, int B, int Mtot, int H, int P) {
    SolverWs w;
    char* p = reinterpret_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off += al(n); return q; };
    w.kn = (double4*)take(sizeof(double4) * (size_t)Mtot);
    w.kp = (double4*)take(sizeof(double4) * (size_t)Mtot);
    w.wq = (uint32_t*)take(sizeof(uint32_t) * (size_t)Mtot);
    w.cdf = (uint32_t*)take(sizeof(uint32_t) * (size_t)Mtot);
    w.pp = (PairPrior*)take(sizeof(PairPrior) * (size_t)B);
    w.tgt = (double*)take(sizeof(double) * (size_t)B * P * 3);
    w.F_all = (double*)take(sizeof(double) * (size_t)B * H * 9);
    w.pscore = (double*)take(sizeof(double) * (size_t)B * H);
    w.count_all = (int*)take(sizeof(int) * (size_t)B * H);
    w.score_all = (double*)take(sizeof(double) * (size_t)B * H);
    w.bits = (uint8_t*)take((size_t)Mtot);
    w.status_sel = (int*)take(sizeof(int) * (size_t)B);
    w.n_inl = (int*)take(sizeof(int) * (size_t)B);
    w.good = (int*)take(sizeof(int) * (size_t)B * 4);
    w.rt = (double*)take(sizeof(double) * (size_t)B * 21);
    w.bytes = off;
    return w;
}

// Stages 1-3 of the solver (prior set-up, normalisation + bias weights, hypotheses, verification), shared by far_solver_f64 and
// far_ransac_f64.  false: a launch-configuration call failed (far_check_launch reports it).
static bool solver_stages_1to3(const float* kpts0, const float* kpts1, const int* offsets, int B, int Mtot, int Mmax, const double* K0,
                               const double* K1, const double* inl_th, const float* priorRT, const float* pcl, int P, double prior_lambda,
                               int H, int minimal, uint32_t seed, const int* samples_in, double* F_all_out, int* count_all_out,
                               double* score_all_out, int* samples_out, const SolverWs& w, hipStream_t stream, double*& F_all,
                               double*& score_all) {
    const bool prior = priorRT != nullptr;
    if (prior)
        hipLaunchKernelGGL(k_prior_setup, dim3(B), dim3(128), 0, stream, priorRT, B, w.pp, pcl, P, w.tgt);
    if (Mtot > 0 && Mmax > 0) {
        hipLaunchKernelGGL(k_prepare, dim3((Mmax + 255) / 256, B), dim3(256), 0, stream, kpts0, kpts1, offsets, K0, K1,
                           prior ? w.pp : nullptr, w.kn, w.kp, w.wq);
        if (prior && !samples_in) hipLaunchKernelGGL(k_cdf, dim3(B), dim3(256), 0, stream, w.wq, offsets, w.cdf);
    }
    F_all = F_all_out ? F_all_out : w.F_all;
    int* count_all = count_all_out ? count_all_out : w.count_all;
    score_all = score_all_out ? score_all_out : w.score_all;
    // hypotheses: the normalized 8-point for pairs with >= 8 correspondences (minimal = 8), the five-point solver for pairs
    // with 5..7 -- and for every pair when minimal = 5.  Explicit samples (tests) follow the mode: [B][H][8] or [B][H/10][5].
    const int* s8 = minimal == 8 ? samples_in : nullptr;
    const int* s5 = minimal == 5 ? samples_in : nullptr;
    if (minimal == 8)
        hipLaunchKernelGGL(k_hypotheses, dim3((H + 63) / 64, B), dim3(64), 72 * 64 * sizeof(double), stream, w.kp, offsets,
                           (prior && !samples_in) ? w.cdf : nullptr, s8, H, minimal, seed, prior ? w.pp : nullptr, pcl, w.tgt,
                           P, prior_lambda, F_all, w.pscore, minimal == 8 ? samples_out : nullptr);
    {
        const int H5 = H / 10 > 0 ? H / 10 : 1;
        constexpr int smem5 = 200 * 64 * sizeof(double);
        bool cfg_failed = false;
        FAR_ONCE_PER_DEVICE(cfg_failed = hipFuncSetAttribute((const void*)k_hypotheses5, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             smem5) != hipSuccess);
        if (cfg_failed) return false;
        hipLaunchKernelGGL(k_hypotheses5, dim3((H5 + 63) / 64, B), dim3(64), smem5, stream, w.kp, offsets,
                           (prior && !samples_in) ? w.cdf : nullptr, s5, H, minimal, seed, prior ? w.pp : nullptr, pcl, w.tgt,
                           P, prior_lambda, F_all, w.pscore, minimal == 5 ? samples_out : nullptr);
    }
    hipLaunchKernelGGL(k_score, dim3((H + 63) / 64, B), dim3(256), 0, stream, w.kp, offsets, F_all, w.pscore, inl_th, H,
                       count_all, score_all);
    return true;
}

}  // namespace

extern "C" {

size_t far_solver_workspace_bytes(int B, int Mtot, int H, int P) { return carve(nullptr, B, Mtot > 0 ? Mtot : 1, H, P > 0 ? P : 1).bytes; }

// Batched pose solver for B pairs (see include/far_hip.h for the full contract).
int far_solver_f64(const float* kpts0, const float* kpts1, const int* offsets, int B, int Mtot, int Mmax,
                   const double* K0, const double* K1, const double* inl_th, int many_thr,
                   const float* priorRT, const float* pcl, int P, double prior_lambda,
                   int H, int minimal, uint32_t seed, const int* samples_in,
                   double* R_out, double* t_out, double* E_out, uint8_t* mask_out, int* status_out,
                   int* num_after_out, int* n_tight_out, int* n_ultra_out, int* n_cheir_out, int* best_out,
                   double* F_all_out, int* count_all_out, double* score_all_out, int* samples_out,
                   void* ws, hipStream_t stream) {
    far_clear_errors();
    if (B <= 0) return FAR_OK;
    if (!offsets || !K0 || !K1 || !inl_th || !R_out || !t_out || !E_out || !status_out ||
        !num_after_out || !n_tight_out || !n_ultra_out || !n_cheir_out || !best_out || !ws || H <= 0 || Mtot < 0 ||
        (minimal != 8 && minimal != 5) || (minimal == 5 && H < 10))
        return FAR_EINVAL;
    // a batch in which no pair has a correspondence (Mtot == 0) is legal: every pair reports status 0 (metrics.py:83-85);
    // the per-correspondence arrays are then empty and may be null
    if (Mtot > 0 && (!kpts0 || !kpts1 || !mask_out)) return FAR_EINVAL;
    if (priorRT && (!pcl || P <= 0)) return FAR_EINVAL;
    SolverWs w = carve(ws, B, Mtot > 0 ? Mtot : 1, H, P > 0 ? P : 1);
    double* F_all = nullptr;
    double* score_all = nullptr;
    if (!solver_stages_1to3(kpts0, kpts1, offsets, B, Mtot, Mmax, K0, K1, inl_th, priorRT, pcl, P, prior_lambda, H, minimal, seed,
                            samples_in, F_all_out, count_all_out, score_all_out, samples_out, w, stream, F_all, score_all))
        return far_check_launch();
    hipLaunchKernelGGL(k_select, dim3(B), dim3(256), 0, stream, w.kp, offsets, F_all, score_all, inl_th, H, many_thr,
                       minimal, best_out, E_out, mask_out, w.n_inl, n_tight_out, n_ultra_out, w.status_sel, w.good, 0);
    const int nsplit = Mmax > 0 ? (Mmax + 255) / 256 : 1;
    hipLaunchKernelGGL(k_recover_bits, dim3(nsplit, B), dim3(256), 0, stream, w.kn, offsets, E_out, w.status_sel, 1e9,
                       (const uint8_t*)mask_out, w.bits, w.good, w.rt);
    hipLaunchKernelGGL(k_recover_final, dim3(B), dim3(256), 0, stream, offsets, w.status_sel, (const int*)w.good, (const double*)w.rt,
                       mask_out, (const uint8_t*)w.bits, R_out, t_out, n_cheir_out, num_after_out, status_out);
    return far_check_launch();
}

// RANSAC.forward (third_party/prior_ransac/ransac.py:340-442) on its own: stages 1-4 of far_solver_f64 WITHOUT recoverPose, for
// correspondences that are already in the coordinates the caller wants the model in (the reference passes K-normalised points,
// metrics.py:124-127): K0 = K1 = identity inside.  Arguments as far_solver_f64; outputs: E_out [B][9] (zeros when no model scored
// above the minimal sample size, as best_model_total stays zeros(3, 3), :354), mask_out [Mtot] with bit 0 = inlier at inl_th,
// bit 1 = at inl_th / 10, bit 2 = at inl_th / 100 (:284-287), the three counts, best_out [B] (-1: none).
int far_ransac_f64(const float* kp1, const float* kp2, const int* offsets, int B, int Mtot, int Mmax, const double* inl_th,
                   const float* priorRT, const float* pcl, int P, double prior_lambda, int H, int minimal, uint32_t seed,
                   const int* samples_in, double* E_out, uint8_t* mask_out, int* n_inl_out, int* n_tight_out, int* n_ultra_out,
                   int* best_out, void* ws, hipStream_t stream) {
    far_clear_errors();
    if (B <= 0) return FAR_OK;
    if (!offsets || !inl_th || !E_out || !n_inl_out || !n_tight_out || !n_ultra_out || !best_out || !ws || H <= 0 || Mtot < 0 ||
        (minimal != 8 && minimal != 5) || (minimal == 5 && H < 10) || (Mtot > 0 && (!kp1 || !kp2 || !mask_out)) ||
        (priorRT && (!pcl || P <= 0)))
        return FAR_EINVAL;
    SolverWs w = carve(ws, B, Mtot > 0 ? Mtot : 1, H, P > 0 ? P : 1);
    // identity intrinsics for k_prepare: [B][9] packed into the workspace's R | t scratch (21 doubles per pair, unused without recoverPose)
    double* Kid = w.rt;
    hipLaunchKernelGGL(k_identity9, dim3((B + 63) / 64), dim3(64), 0, stream, Kid, B);
    double* F_all = nullptr;
    double* score_all = nullptr;
    if (!solver_stages_1to3(kp1, kp2, offsets, B, Mtot, Mmax, Kid, Kid, inl_th, priorRT, pcl, P, prior_lambda, H, minimal, seed,
                            samples_in, nullptr, nullptr, nullptr, nullptr, w, stream, F_all, score_all))
        return far_check_launch();
    hipLaunchKernelGGL(k_select, dim3(B), dim3(256), 0, stream, w.kp, offsets, F_all, score_all, inl_th, H, 1, minimal, best_out, E_out,
                       mask_out, n_inl_out, n_tight_out, n_ultra_out, w.status_sel, w.good, 1);
    return far_check_launch();
}

// run_8point(points1, points2, weights) (cv_geometry.py:772-833): B problems x N >= 8 correspondences, float64 in and out.
// p1, p2 [B][N][2]; w [B][N] or NULL (ones); F_out [B][9] = normalize_transformation(T2^T F_rank2 T1).
int far_eightpoint_f64(const double* p1, const double* p2, const double* w, int B, int N, double* F_out, hipStream_t stream) {
    far_clear_errors();
    if (B <= 0) return FAR_OK;
    if (!p1 || !p2 || !F_out || N < 8) return FAR_EINVAL;
    hipLaunchKernelGGL(k_eightpoint, dim3((B + 63) / 64), dim3(64), 72 * 64 * sizeof(double), stream, p1, p2, w, B, N, F_out);
    return far_check_launch();
}

// decompose_essential_matrix(E) (essential.py:99-139): n matrices [n][9] -> R1, R2 [n][9], t [n][3], float64.
int far_decompose_essential_f64(const double* E, long n, double* R1_out, double* R2_out, double* t_out, hipStream_t stream) {
    far_clear_errors();
    if (n <= 0) return FAR_OK;
    if (!E || !R1_out || !R2_out || !t_out) return FAR_EINVAL;
    hipLaunchKernelGGL(k_decompose, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, E, n, R1_out, R2_out, t_out);
    return far_check_launch();
}

}  // extern "C"
