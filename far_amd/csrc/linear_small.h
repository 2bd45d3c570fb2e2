// Internal interface between K9's entry point (conv_igemm_f16s.hip) and the few-row Linear kernel (linear_small_f16s.hip).
#pragma once
#include "common.h"

struct LinSmallArgs {
    const float* x;              // [rows][Cin] fp32
    const unsigned char* w;      // K9's packed split-fp16 image of the weight (far_conv_pack_*), ksize 1
    const float* scale;          // [Cout]
    const float* shift;          // [Cout] or null
    const float* res;            // y's layout, or null
    float* y;
    long rows;
    int Cin, Cout, Csub, NT, nblkY, act;
    float slope, act_scale, out_mul;
    const float* scale_dev;      // device { act_scale, out_mul } or null
    int* overflow;
};

// true when the kernel covers the launch (the caller then must not launch K9)
bool far_linear_small_covers(long rows, int Cin, int Cout);
int far_linear_small_launch(const LinSmallArgs& a, hipStream_t stream);
