// K20: AdamW for all parameters of a model in ONE launch (training step of BASELINE configs[2]).
//
// Replaces the multi-tensor kernels of torch.optim.AdamW (mp3d_loftr/src/optimizers/__init__.py:5-16 builds it; ~30 launches per
// step for the 189 parameter tensors) with the same arithmetic in the same order, element by element:
//     p   *= 1 - lr wd
//     m    = m + (1 - beta1) (g - m)                               (torch's lerp)
//     v    = beta2 v + (1 - beta2) g g
//     p   -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)             bc_i = 1 - beta_i^step
// A device table holds one row per tensor {p, g, m, v, n} and one {tensor, chunk} pair per workgroup (4096 elements each); the
// gradient pointers are refreshed every step (autograd re-allocates them), the rest is built once.
#include "common.h"

namespace {

struct AdamRow {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
};

constexpr int ADAM_CHUNK = 4096;

__global__ __launch_bounds__(256) void k_adamw(const AdamRow* __restrict__ rows, const int2* __restrict__ blocks, float decay, float w1,
                                               float beta2, float w2, float step, float bc2_sqrt, float eps) {
    const int2 b = blocks[blockIdx.x];
    const AdamRow r = rows[b.x];
    if (!r.g) return;                                   // a parameter without a gradient this step is left alone (as torch does)
    const long base = (long)b.y * ADAM_CHUNK;
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const long i = base + k * 256 + threadIdx.x;
        if (i >= r.n) break;
        const float g = r.g[i];
        float p = r.p[i] * decay;
        float m = r.m[i];
        m = m + w1 * (g - m);
        float v = r.v[i] * beta2;
        v = v + (w2 * g) * g;                           // addcmul: value * tensor1 * tensor2, left to right
        const float denom = __builtin_sqrtf(v) / bc2_sqrt + eps;
        p = p - step * (m / denom);
        r.p[i] = p; r.m[i] = m; r.v[i] = v;
    }
}

}  // namespace

extern "C" {

// Bytes of the device table for n tensors with `nblocks` = sum over tensors of ceil(numel / 4096) workgroups.
long far_adamw_table_bytes(int n, long nblocks) {
    if (n <= 0 || nblocks <= 0) return 0;
    return (long)n * (long)sizeof(AdamRow) + nblocks * (long)sizeof(int2);
}

// One AdamW step over the table: rows = table, blocks = table + n rows.  bc1 = 1 - beta1^step, bc2_sqrt = sqrt(1 - beta2^step).
// Hyper-parameters in double: the derived scalars (1 - lr wd, 1 - beta, lr / bc1) are formed in double and rounded once, as torch does.
int far_adamw_step_f32(const void* table, int n, long nblocks, double lr, double beta1, double beta2, double eps, double wd, double bc1,
                       double bc2_sqrt, hipStream_t stream) {
    far_clear_errors();
    if (!table || n <= 0 || nblocks <= 0 || nblocks > 0x7fffffffL || !(bc1 > 0.0) || !(bc2_sqrt > 0.0)) return FAR_EINVAL;
    const AdamRow* rows = reinterpret_cast<const AdamRow*>(table);
    const int2* blocks = reinterpret_cast<const int2*>(rows + n);
    // the derived scalars in double, rounded once, as torch passes them to its kernels
    hipLaunchKernelGGL(k_adamw, dim3((unsigned)nblocks), dim3(256), 0, stream, rows, blocks, (float)(1.0 - lr * wd), (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1), (float)bc2_sqrt, (float)eps);
    return far_check_launch();
}

}  // extern "C"
