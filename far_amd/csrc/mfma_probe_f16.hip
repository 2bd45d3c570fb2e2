// far_mfma_probe_f16: what the f16 matrix pipe of THIS part sustains, measured with the register / LDS footprint of
// the split-precision kernels (K9, K1, K2) and nothing else in the loop.  bench.py times it on the box it runs on and
// reports it as roofline.sustained_peak next to the nominal 2.5 PFLOP/s: under dense v_mfma_f32_32x32x16_f16 issue the
// shader clock of an MI355X is power-limited (1.5-1.8 GHz against the 2.4 GHz the nominal peak assumes), so the nominal
// figure is not reachable by any kernel; this one bounds what is.
//
// Shape of the loop = K9's phase body: a wave owns 2 x 4 accumulator tiles (128 accumulator registers), per k-step it
// holds 2 A fragments and 4 B fragments (hi parts; mode 1 re-reads them from LDS every step as K9 does, 6 ds_read_b128
// per 8 MFMAs; mode 0 keeps them in registers) and issues 8 MFMAs back to back.  Operands are pseudo-random fp16 values
// in [-1, 1): multiplying zeros draws measurably less power and clocks higher.  4 waves per workgroup, 2 workgroups per
// CU (the occupancy of K9 / K1 / K2), grid = 2 x CU count x `rounds`.
#include "common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_mfma_probe(int iters, int lane_step, unsigned seed, float* sink) {
    __shared__ h8 frag[6][4][64];                   // 6 fragments x 4 waves x 64 lanes x 16 B = 24 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    h8 a[2], b[4];
    auto rnd = [&]() {
        h8 v;
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            v[e] = (_Float16)(((int)(s >> 8) & 0xffff) * (1.0f / 32768.0f) - 1.0f);
        }
        return v;
    };
    for (int i = 0; i < 2; ++i) a[i] = rnd();
    for (int i = 0; i < 4; ++i) b[i] = rnd();
    if (MODE == 1) {
        for (int i = 0; i < 2; ++i) frag[i][wave][lane] = a[i];
        for (int i = 0; i < 4; ++i) frag[2 + i][wave][lane] = b[i];
        __syncthreads();
    }
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            // lane_step is 0 at run time; the compiler cannot know, so the six reads stay in the loop
            const int l2 = (lane + it * lane_step) & 63;
            for (int i = 0; i < 2; ++i) a[i] = frag[i][wave][l2];
            for (int i = 0; i < 4; ++i) b[i] = frag[2 + i][wave][l2];
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 12345.678f) sink[0] = t;               // keeps the accumulators alive; never true for these operands
}

}  // namespace

// Launches the probe: `rounds` x (2 workgroups per CU) workgroups of 4 waves, each wave `iters` steps of 8 MFMAs.
// *flops_out (host) receives the executed flop count of the launch: waves x iters x 8 x (2 * 32 * 32 * 16).
// mode 0: operands in registers; mode 1: operand fragments re-read from LDS every step (K9's footprint).
extern "C" int far_mfma_probe_f16(int mode, int iters, int rounds, float* sink, double* flops_out, hipStream_t stream) {
    far_clear_errors();
    if (iters <= 0 || rounds <= 0 || !sink || (mode != 0 && mode != 1)) return FAR_EINVAL;
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int blocks = 2 * cus * rounds;
    if (mode == 0) k_mfma_probe<0><<<blocks, 256, 0, stream>>>(iters, 0, 0x9e3779b9u, sink);
    else k_mfma_probe<1><<<blocks, 256, 0, stream>>>(iters, 0, 0x9e3779b9u, sink);
    if (flops_out) *flops_out = (double)blocks * 4.0 * (double)iters * 8.0 * (2.0 * 32 * 32 * 16);
    return far_check_launch();
}
