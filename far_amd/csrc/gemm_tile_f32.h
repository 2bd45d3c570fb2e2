// LDS-tiled fp32 correlation tile engine on the exact-f32 matrix core path
// (v_mfma_f32_32x32x2_f32; bitwise an fmaf chain, see cdna_hip_programming.md section 3).
//
// Workgroup = 256 threads = 4 waves.  WG tile = 128 "row" tokens x 128 "column" tokens.
// Wave w owns row tokens [32w, 32w+32) against all 128 column tokens (4 MFMA tiles of 32x32,
// 64 accumulator registers).  The channel dimension is streamed in chunks of 32 floats through a
// double-buffered, padded LDS image ([128][36] floats per operand: row stride 144 B = 9 16-B slots,
// odd, so the 16-lane groups of ds_read_b128 hit 16 distinct slots -> conflict free).
//
// Each lane's MFMA k-assignment: lane half h = lane>>5 supplies channels {8q+4h .. 8q+4h+3} of every
// 8-channel group q, one component per MFMA; A and B agree, so the permutation of k is harmless.
#pragma once
#include "common.h"

#define TILE_M 128
#define TILE_N 128
#define KC 32
#define LDS_STRIDE 36  // floats per LDS row (32 + 4 pad)

struct TileLds {
    float a[2][TILE_M * LDS_STRIDE];
    float b[2][TILE_N * LDS_STRIDE];
};

// Global -> register staging of one 128x32 chunk of A and of B (4 float4 each per thread).
struct ChunkRegs {
    float4 a[4];
    float4 b[4];
};

__device__ __forceinline__ float4 ldg4_guard(const float* __restrict__ base, int row, int nrows,
                                            int ld, int col) {
    if (row < nrows) return *reinterpret_cast<const float4*>(base + (size_t)row * ld + col);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void chunk_load(ChunkRegs& r, const float* __restrict__ A, int a_row0,
                                           int a_rows, const float* __restrict__ B, int b_row0,
                                           int b_rows, int ld, int k0, int tid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int idx = tid + 256 * p;
        int row = idx >> 3, slot = idx & 7;
        r.a[p] = ldg4_guard(A, a_row0 + row, a_rows, ld, k0 + slot * 4);
        r.b[p] = ldg4_guard(B, b_row0 + row, b_rows, ld, k0 + slot * 4);
    }
}

__device__ __forceinline__ float4 div4(float4 v, float d) {
    return make_float4(v.x / d, v.y / d, v.z / d, v.w / d);
}

// feat_div: every feature element is divided by it when staged (the reference divides both
// feature maps by sqrt(C) before the contraction: coarse_matching.py:104-105).  Pass 1.0f for none.
__device__ __forceinline__ void chunk_store(const ChunkRegs& r, TileLds& lds, int buf, int tid,
                                            float feat_div) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int idx = tid + 256 * p;
        int row = idx >> 3, slot = idx & 7;
        float4 va = r.a[p], vb = r.b[p];
        if (feat_div != 1.0f) {
            va = div4(va, feat_div);
            vb = div4(vb, feat_div);
        }
        *reinterpret_cast<float4*>(&lds.a[buf][row * LDS_STRIDE + slot * 4]) = va;
        *reinterpret_cast<float4*>(&lds.b[buf][row * LDS_STRIDE + slot * 4]) = vb;
    }
}

// acc[ct] (+)= tile product for one 32-channel chunk: D[m = row token][n = col token]; a lane holds column token
// 32ct + (lane & 31) and row tokens 32w + mfma32_row(reg, h).
__device__ __forceinline__ void chunk_mfma(f32x16 (&acc)[4], const TileLds& lds, int buf, int wave,
                                           int lane) {
    const int l31 = lane & 31, h = lane >> 5;
    const float* arow = &lds.a[buf][(32 * wave + l31) * LDS_STRIDE + 4 * h];
    const float* brow = &lds.b[buf][l31 * LDS_STRIDE + 4 * h];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 av = *reinterpret_cast<const f32x4*>(arow + 8 * q);
        f32x4 bv[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            bv[ct] = *reinterpret_cast<const f32x4*>(brow + ct * 32 * LDS_STRIDE + 8 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[ct][c], acc[ct], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ void acc_zero(f32x16 (&acc)[4]) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
}
