// Shared device helpers for the far_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <atomic>

#define FAR_OK 0
#define FAR_EINVAL (-22)
#define FAR_ELAUNCH (-5)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// The fp16 (hi, lo) split of two values at once: hi = fp16(x), lo = fp16(x - hi), round to nearest.  Round 3: the packed instructions
// of gfx950 (v_cvt_pk_f16_f32, v_cvt_f32_f16 x2, v_pk_add_f32, v_cvt_pk_f16_f32: 5 per pair instead of the 8 scalar ones); round 4:
// v_cvt_pk_f16_f32 + one v_fma_mix per lo half (3 per pair) -- the same bits (K17 found and measured it first: 3.00 vs 3.12 ms;
// the whole step 93.35 -> 93.05 ms, the training step 38.0 -> 37.7 ms).  -DFAR_SPLIT2_PACKED selects the round-3 form.
__device__ __forceinline__ void split2(f32x2 x, f16x2& hi, f16x2& lo) {
    hi = __builtin_convertvector(x, f16x2);
#ifdef FAR_SPLIT2_PACKED
    lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x2), f16x2);
#else
    // round 4: one v_fma_mix per lo half -- the fp16 operand is widened, subtracted from x in fp32 (exactly) and the result rounded to
    // fp16 by the same instruction: the same bits as the packed form above in 3 instructions per pair instead of 5
    const unsigned h = __builtin_bit_cast(unsigned, hi);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x.x));
    // The trailing s_nop is part of the instruction's contract here: v_fma_mixhi_f16 writes HALF a register (op_sel hi), and on gfx950 a
    // matrix instruction that reads such a register too soon still sees the old upper half (the destination-select forwarding hazard;
    // hipcc pads it for instructions it knows, an asm statement it cannot see into).  Round 6 found it as K9's k | v state epilogue
    // returning different sums from launch to launch in one particular build (docs/rounds/r06.md section 2g): `mixhi, <one instruction>,
    // mfma` reads the stale half, two instructions in between do not.  Two wait states per pair, < 1 % of any kernel's VALU time;
    // far_amd/build.py checks the distance to the first VALU / MFMA reader on the generated code.
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(l) : "v"(h), "v"(x.y));
    lo = __builtin_bit_cast(f16x2, l);
#endif
}

// Row index inside a 32x32 MFMA accumulator tile for accumulator register `r` (0..15) of a lane
// whose upper-half flag is `h` (lane >> 5).  Column index is lane & 31.
// (cdna_hip_programming.md section 3: row=(reg&3)+8*(reg>>2)+4*(lane>>5))
__device__ __forceinline__ int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float shfl_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ int shfl_xor_i(int v, int m) { return __shfl_xor(v, m, 64); }

// exp(x) on the transcendental unit: v_exp_f32(x * log2(e)).  For the arguments used here (x <= 0 after
// max subtraction) the absolute error of e^x is <= |x| e^x 2^-24 + 1 ulp <= ~3e-8, far below the 1e-5
// parity tolerance on confidences; a denormal result flushes to 0, as irrelevant.
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// Two workgroups of an MFMA-bound kernel share each SIMD (one wave each).  Left alone they drift into lockstep:
// both waves sit in their MFMA phase together (pipe shared, each at half speed), then both in their VALU/LDS tail
// together (pipe idle).  Giving the wave in the even hardware slot a higher issue priority makes the pipe go to
// one wave at a time, so one wave's tail overlaps the other's MFMAs.  Speed only; correctness never depends on it.
__device__ __forceinline__ void stagger_priority_by_wave_slot(int enable) {
    if (!enable) return;
    // HW_REG_HW_ID (id 4): WAVE_ID in bits [3:0]
    unsigned hwid = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);
    if ((hwid & 1u) == 0) __builtin_amdgcn_s_setprio(2);
}

// XCD-aware block -> (problem z, row block Ib) mapping for the tiled correlation kernels (1-D grid of nI * Z blocks).
// Hardware places block b on XCD b % 8 (observed, speed only).  All nI row blocks of one problem re-read the same
// column-side operand, so they are steered to ONE XCD's L2: measured HBM/fabric fetch of k_match_f32 at batch 32 was
// 22x the algorithmic bytes with the naive (Ib fastest) order.  Falls back to the plain order when Z % 8 != 0.
__device__ __forceinline__ void tile_coords(int nI, int Z, int& z, int& Ib) {
    const int id = blockIdx.x;
    if ((Z & 7) == 0) {
        const int xcd = id & 7, li = id >> 3;
        z = xcd + 8 * (li / nI);
        Ib = li % nI;
    } else {
        z = id / nI;
        Ib = id - z * nI;
    }
}

// x / d for a loop-invariant divisor d, given r = 1.0f / d (IEEE).  q = fl(x r); e = fma(-d, q, x); q' = fma(e, r, q)
// is the correctly rounded quotient (Markstein) except for divisors whose significand is all ones; 3 instructions
// instead of the ~10 of the generic v_div_* sequence.
__device__ __forceinline__ float fdiv_by(float x, float d, float r) {
    float q = x * r;
    float e = fmaf(-d, q, x);
    return fmaf(e, r, q);
}

// Merge two online-softmax partials (m, s): s = sum exp(x - m).  Safe for empty partials
// (m = -FLT_MAX, s = 0).
__device__ __forceinline__ void softmax_merge(float& m, float& s, float mo, float so) {
    float mn = fmaxf(m, mo);
    float a = (m == mn) ? 1.0f : fexp(m - mn);
    float b = (mo == mn) ? 1.0f : fexp(mo - mn);
    s = s * a + so * b;
    m = mn;
}

// The barrier of a hand-synchronised LDS-DMA ring (K9, K13, K14, K17: asm global_load_lds_dwordx4 into a slot right behind the
// barrier that follows the slot's last reader).  RULE: no LDS read of the wave may be outstanding when it arrives -- nothing orders
// an LDS-DMA write behind a ds_read that is still queued, and `asm volatile("s_barrier")` pins memory instructions only: hipcc
// sinks the last MFMAs of a phase and the s_waitcnt lgkmcnt in front of them below it, so without this explicit wait a wave crosses
// the barrier with its last fragment reads still in the LDS queue and a sibling's (L2-hot) re-request of the slot can land first.
// That was the run-to-run difference of K14's two-workgroups-per-CU forms in round 5 (single wrong windows; root cause and the
// ring-only reproducer: docs/rounds/r06.md section 1, tools/ubench/ring_war.hip).  far_amd/build.py scans the generated code of every
// kernel that issues LDS-DMA for a barrier with LDS reads outstanding and fails the build on one.
// LGKM = false exists for the FAR_RING_EXP experiment builds only (the rounds-3..5 form, to reproduce the race).
template <bool LGKM = true>
__device__ __forceinline__ void ring_barrier() {
    if (LGKM) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_barrier" ::: "memory");
}

// Vendor libraries (rocBLAS / hipBLASLt kernel lookups) can leave a benign error in HIP's per-thread "last error"
// slot; every entry point clears it first so that far_check_launch() only reports this library's own launches.
static inline void far_clear_errors() { (void)hipGetLastError(); }

extern "C" int far_last_hip_error(void);
int far_get_tuning(int key);
void far_record_hip_error(int e);

// Kernel attributes (hipFuncAttributeMaxDynamicSharedMemorySize) are PER DEVICE: a process-wide `static bool` would
// leave every GPU but the first unconfigured (a gather on another device, a test that switches cuda:1).  One bit per
// device ordinal per call site; the bit is published after the (idempotent) configuration call has returned, so a
// thread that sees it set may launch, and two racing threads at worst both configure.
static inline int far_current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 63;
}
#define FAR_ONCE_PER_DEVICE(...)                                                        \
    do {                                                                                \
        static std::atomic<unsigned long long> far_done_{0};                            \
        const unsigned long long far_bit_ = 1ull << far_current_device();               \
        if (!(far_done_.load(std::memory_order_acquire) & far_bit_)) {                  \
            __VA_ARGS__;                                                                \
            far_done_.fetch_or(far_bit_, std::memory_order_release);                    \
        }                                                                               \
    } while (0)

static inline int far_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) far_record_hip_error((int)e);
    return e == hipSuccess ? FAR_OK : FAR_ELAUNCH;
}
