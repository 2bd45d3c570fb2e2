// K2: the bilinear dual-softmax attention of the EMM head, without materialising the N x N scores.
//
// Replaces (reference, mp3d_loftr/src/loftr/loftr_module/transformer.py, CrossAttention.forward):
//   :275-276  attn = (q @ k^T) * scale
//   :281-282  attn_fundamental = attn.softmax(-1) * attn.softmax(-2)
//   :284-286  v~ = cat([v, positional(6)], dim=3)                       (B, h, N, 70)
//   :291-292  fundamental = (v~^T @ attn_fundamental) @ v~               (B, h, 70, 70)
//
// Algebra used here:  F = v~^T (P v~).  This kernel produces T = P v~  (Z, N, 70) in one streaming pass
// over the score tiles, given the softmax statistics from far_dual_softmax_stats_f32; the trailing
// 70 x N x 70 contraction is a plain small GEMM done by the caller (rocBLAS through torch.bmm).
//
// Tile engine: WG = 4 waves, 128 query tokens (i) x 128 key tokens (j) per step.  The score tile is
// computed TRANSPOSED on the f32 matrix core (D[m=j][n=i]) so that each lane ends up holding, for its
// query row i = lane&31, sixteen P values whose j-order is exactly the k-order the next MFMA wants for its
// A operand: P feeds the second contraction straight from the accumulator registers (no LDS round trip).
// The query panel (32 rows x 64 channels per wave) lives in registers for the whole sweep.
#include "common.h"

namespace {

constexpr int EM_D = 64;       // head dim (channels of q / k / v)
constexpr int EM_DV = 70;      // v~ width (64 + 6 positional)
constexpr int EM_KS = 68;      // LDS row stride of the key tile   (floats; 17 16-B slots: odd)
constexpr int EM_VS = 72;      // LDS row stride of the v~ tile    (floats)

struct EmmLds {
    float k[128 * EM_KS];
    float v[128 * EM_VS];
    float2 cst[128];
};

__global__ __launch_bounds__(256, 2) void k_emm_pv_f32(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ pos,  // [N][6]
    int Z, int N, float scale, const float2* __restrict__ rowstat, const float2* __restrict__ colstat,
    float* __restrict__ T,          // [Z][N][70]
    int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    EmmLds& lds = *reinterpret_cast<EmmLds*>(smem_raw);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    int z, Ib;
    tile_coords((N + 127) / 128, Z, z, Ib);
    const int i0 = Ib * 128;
    const float* Q = q + (size_t)z * N * EM_D;
    const float* K = k + (size_t)z * N * EM_D;
    const float* V = v + (size_t)z * N * EM_D;

    // query fragment: row i = i0 + 32*wave + l31, channels {8g + 4h .. +3}, g = 0..7
    const int irow = i0 + 32 * wave + l31;
    const bool ivalid = irow < N;
    f32x4 qa[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (ivalid) qa[g] = *reinterpret_cast<const f32x4*>(Q + (size_t)irow * EM_D + 8 * g + 4 * h);
        else qa[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float2 rst = ivalid ? rowstat[(size_t)z * N + irow] : make_float2(0.f, 1.f);
    const float rinv = 1.0f / rst.y;

    stagger_priority_by_wave_slot(stagger);
    // T[:, 0:64] accumulates on the matrix core (2 tiles); the 6 positional columns T[:, 64:70] are accumulated
    // on the VALU in the MFMA shadow: in this orientation a lane owns ONE query row, so they cost 6 registers,
    // and a third (mostly padding) MFMA tile is avoided.
    f32x16 tacc[2];
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int r = 0; r < 16; ++r) tacc[bt][r] = 0.f;
    float tpos[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int nJ = (N + 127) / 128;
    for (int Jt = 0; Jt < nJ; ++Jt) {
        const int j0 = Jt * 128;
        __syncthreads();  // previous tile fully consumed
        // stage key tile [128][64] and v~ tile [128][70 (+2 pad)]
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            int idx = tid + 256 * p;       // 0..2047
            int row = idx >> 4, slot = idx & 15;
            int j = j0 + row;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (j < N) {
                kv = *reinterpret_cast<const float4*>(K + (size_t)j * EM_D + slot * 4);
                vv = *reinterpret_cast<const float4*>(V + (size_t)j * EM_D + slot * 4);
            }
            *reinterpret_cast<float4*>(&lds.k[row * EM_KS + slot * 4]) = kv;
            *reinterpret_cast<float4*>(&lds.v[row * EM_VS + slot * 4]) = vv;
        }
        {
            // positional columns 64..71 (70, 71 are zero padding)
            int row = tid >> 1, half = tid & 1;  // 128 rows x 2 float4
            int j = j0 + row;
            float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < N) {
                const float* pp = pos + (size_t)j * 6;
                if (half == 0) pv = make_float4(pp[0], pp[1], pp[2], pp[3]);
                else pv = make_float4(pp[4], pp[5], 0.f, 0.f);
            }
            *reinterpret_cast<float4*>(&lds.v[row * EM_VS + 64 + 4 * half]) = pv;
            if (tid < 128) {
                int jj = j0 + tid;
                float2 c = jj < N ? colstat[(size_t)z * N + jj] : make_float2(0.f, 1.f);
                lds.cst[tid] = make_float2(c.x, 1.0f / c.y);   // (max, 1 / sum)
            }
        }
        __syncthreads();

        // The 128-key tile is processed as two 64-key halves so that only two score accumulators (32 registers)
        // are live at a time; two independent MFMA chains are exactly enough to cover the 64-cycle
        // dependent-accumulator latency of v_mfma_f32_32x32x2_f32.
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int jb = 64 * half;                                  // first key row of this half inside the tile
            // ---- scores, transposed: D[m = j][n = i] ----
            f32x16 acc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
            const float* krow = &lds.k[(jb + l31) * EM_KS + 4 * h];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f32x4 kb[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) kb[ct] = *reinterpret_cast<const f32x4*>(krow + ct * 32 * EM_KS + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[ct][c], qa[g][c], acc[ct], 0, 0, 0);
            }

            // ---- P in place: lane holds i = l31, j = j0 + jb + 32ct + mfma32_row(r, h) ----
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jl = jb + 32 * ct + mfma32_row(r, h);
                    const float2 cs = lds.cst[jl];
                    float s = acc[ct][r] * scale;                       // transformer.py:275
                    float pr = fexp(s - rst.x) * rinv;                  // softmax(dim=-1): over keys j
                    float pc = fexp(s - cs.x) * cs.y;                   // softmax(dim=-2): over queries i (cs.y = 1/sum)
                    float p = pr * pc;                                  // :281
                    acc[ct][r] = (ivalid && (j0 + jl) < N) ? p : 0.f;
                }

            // ---- T[i][b] += sum_j P[i][j] v~[j][b]: A = P (from registers), B = v~ tile from LDS ----
            // MFMA step (ct, t): k-pair = { j = 32ct + mfma32_row(t, 0), j = 32ct + mfma32_row(t, 1) }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int jl = jb + 32 * ct + mfma32_row(t, h);
                    const float* vrow = &lds.v[jl * EM_VS];
                    const float b0 = vrow[l31];
                    const float b1 = vrow[32 + l31];
                    const float pj = acc[ct][t];
                    tacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pj, b0, tacc[0], 0, 0, 0);
                    tacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pj, b1, tacc[1], 0, 0, 0);
                    const f32x4 pa = *reinterpret_cast<const f32x4*>(vrow + 64);  // broadcast reads (same j per half-wave)
                    const float2 pb = *reinterpret_cast<const float2*>(vrow + 68);
                    tpos[0] = fmaf(pj, pa[0], tpos[0]);
                    tpos[1] = fmaf(pj, pa[1], tpos[1]);
                    tpos[2] = fmaf(pj, pa[2], tpos[2]);
                    tpos[3] = fmaf(pj, pa[3], tpos[3]);
                    tpos[4] = fmaf(pj, pb.x, tpos[4]);
                    tpos[5] = fmaf(pj, pb.y, tpos[5]);
                }
        }
    }

    // ---- store T: lane holds column b = 32bt + l31, rows i = i0 + 32 wave + mfma32_row(r, h) ----
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
        int b = 32 * bt + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int i = i0 + 32 * wave + mfma32_row(r, h);
            if (i < N) T[((size_t)z * N + i) * EM_DV + b] = tacc[bt][r];
        }
    }
    // positional columns: this lane summed its half (h) of the keys for query row irow
#pragma unroll
    for (int c = 0; c < 6; ++c) tpos[c] += shfl_xor_f(tpos[c], 32);
    if (h == 0 && ivalid) {
#pragma unroll
        for (int c = 0; c < 6; ++c) T[((size_t)z * N + irow) * EM_DV + 64 + c] = tpos[c];
    }
}

}  // namespace

extern "C" {

// T[z] = P[z] @ [v[z] | pos]   with P = softmax_row(s) * softmax_col(s), s = (q k^T) * scale.
//   q, k, v [Z][N][64] fp32 contiguous (q = "query" side that indexes rows of the score matrix)
//   pos [N][6] fp32 (shared by all z); rowstat [Z][N][2], colstat [Z][N][2] from far_dual_softmax_stats_f32
//   (feat_div = 1, sim_div = 1, sim_mul = scale);  T_out [Z][N][70].
int far_emm_pv_f32(const float* q, const float* k, const float* v, const float* pos, int Z, int N, int D,
                   float scale, const float* rowstat, const float* colstat, float* T_out, hipStream_t stream) {
    far_clear_errors();
    if (!q || !k || !v || !pos || !rowstat || !colstat || !T_out || Z <= 0 || N <= 0 || D != EM_D) return FAR_EINVAL;
    FAR_ONCE_PER_DEVICE(
        hipFuncSetAttribute((const void*)k_emm_pv_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(EmmLds)));
    hipLaunchKernelGGL(k_emm_pv_f32, dim3(((N + 127) / 128) * Z), dim3(256), sizeof(EmmLds), stream, q, k, v, pos, Z, N,
                       scale, (const float2*)rowstat, (const float2*)colstat, T_out, (far_get_tuning(0) >> 2) & 1);
    return far_check_launch();
}

}  // extern "C"
