// ABI bookkeeping for libfar_hip.so.
#include "common.h"

static thread_local int g_last_hip_error = 0;
void far_record_hip_error(int e) { g_last_hip_error = e; }

extern "C" int far_abi_version(void) { return 2; }
// hipError_t of the most recent failed launch on this thread (0 = none); for diagnostics after a -5 return.
extern "C" int far_last_hip_error(void) { return g_last_hip_error; }

// Tuning knobs for A/B experiments (speed only; never change results).  key 0: bit mask of kernels that use
// wave-slot priority staggering (1 = k_stats, 2 = k_match, 4 = k_emm_pv).
// This is the library's ONLY process-global state (declared as such in include/far_hip.h); atomics, so that a tuning
// call from one thread is well-defined against launches on another.
static std::atomic<int> g_tuning[8] = {{3}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
extern "C" int far_set_tuning(int key, int value) {
    if (key < 0 || key >= 8) return FAR_EINVAL;
    g_tuning[key].store(value, std::memory_order_relaxed);
    return FAR_OK;
}
int far_get_tuning(int key) { return (key >= 0 && key < 8) ? g_tuning[key].load(std::memory_order_relaxed) : 0; }
