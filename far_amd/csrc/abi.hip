// ABI bookkeeping for libfar_hip.so.
#include "common.h"

static thread_local int g_last_hip_error = 0;
void far_record_hip_error(int e) { g_last_hip_error = e; }

extern "C" int far_abi_version(void) { return 7; }
// The id of the sources this library was built from: sha256/16 over far_amd/csrc/* and the compiler flags (far_amd/build.py
// source_id(), passed as -DFAR_BUILD_ID when this file is compiled).  "unknown" for a build that did not go through build.py.
#ifndef FAR_BUILD_ID
#define FAR_BUILD_ID "unknown"
#endif
extern "C" const char* far_build_id(void) { return FAR_BUILD_ID; }
// hipError_t of the most recent failed launch on this thread (0 = none); for diagnostics after a -5 return.
extern "C" int far_last_hip_error(void) { return g_last_hip_error; }

// Tuning knobs for A/B experiments (speed only; never change results).  key 0: bit mask of kernels that use
// wave-slot priority staggering (1 = k_stats, 2 = k_match, 4 = k_emm_pv); 1: K1 f32 tile variant; 2, 3: conf_matrix writer
// variants; 4: 1 = K9 without the seven-tile mode / K5 windows on the generic path; 5: K5 apply tiles per unit; 6: K5 tokens per
// KV chunk; 7: 1 = K9 Linear launches always on full-height tiles; 8: 1 = K17 (Winograd) splits its operands with the five-instruction
// split2 instead of v_fma_mix (same values); 9: 1 = K17 runs a short last channel block on the full body; 10: 1 = K1's match pass without the tile prescreen; 12: 1 = K10's inference form on the exact-f32 matrix instruction.  13: 1 = K9's FPN merge epilogue in its generic form everywhere; 15: 1 = K17 without the raised issue priority of its multiplying wave group.
// Process-global state of the library: these knobs (atomics, so that a tuning call from one thread is well-defined against
// launches on another) and the per-device one-time kernel attribute setup; the side streams below are per host thread.
static std::atomic<int> g_tuning[16] = {{3}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
extern "C" int far_set_tuning(int key, int value) {
    if (key < 0 || key >= 16) return FAR_EINVAL;
    g_tuning[key].store(value, std::memory_order_relaxed);
    return FAR_OK;
}
int far_get_tuning(int key) { return (key >= 0 && key < 16) ? g_tuning[key].load(std::memory_order_relaxed) : 0; }

// ---- side streams: launches that do not depend on each other (the weight gradient of a layer next to its input gradient; the
// q, k, v projections of a layer) overlap when they are issued on different streams, and at batch 1 each of them fills a
// fraction of the CUs.  Every HOST THREAD keeps FAR_SIDE_STREAMS non-blocking streams (and their fork / join events) per device,
// created on first use: two threads driving one device (a loader thread running inference next to
// the training thread) never record into each other's events -- a side stream then waited on the other thread's record and lost
// the dependency on its own main stream.
//   far_stream_fork(main, i): side stream i waits for everything issued on `main` so far; returns the side stream.
//   far_stream_join(main, i): `main` waits for everything issued on side stream i so far.
// Buffers used on a side stream must stay allocated until the join (the caller's allocator knows only `main`).
#define FAR_SIDE_STREAMS 4
namespace {
struct SideStreams {
    hipStream_t s[FAR_SIDE_STREAMS];
    hipEvent_t fork[FAR_SIDE_STREAMS], join[FAR_SIDE_STREAMS];
    bool ready = false;
};
struct ThreadSides {
    SideStreams dev[64];          // (never destroyed: a thread's exit may come after the HIP runtime's own teardown, and a process has
};                                //  a handful of threads that launch; twelve handles per (thread, device) stay with the runtime)
thread_local ThreadSides t_sides;
SideStreams* side_streams() {
    SideStreams& d = t_sides.dev[far_current_device()];
    if (d.ready) return &d;
    bool ok = true;
    for (int i = 0; i < FAR_SIDE_STREAMS; ++i) {
        ok = ok && hipStreamCreateWithFlags(&d.s[i], hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&d.fork[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&d.join[i], hipEventDisableTiming) == hipSuccess;
    }
    d.ready = ok;
    return ok ? &d : nullptr;
}
}  // namespace

extern "C" void* far_stream_fork(hipStream_t main, int i) {
    SideStreams* d = side_streams();
    if (!d || i < 0 || i >= FAR_SIDE_STREAMS) return nullptr;
    if (hipEventRecord(d->fork[i], main) != hipSuccess || hipStreamWaitEvent(d->s[i], d->fork[i], 0) != hipSuccess) return nullptr;
    return (void*)d->s[i];
}

extern "C" int far_stream_join(hipStream_t main, int i) {
    SideStreams* d = side_streams();
    if (!d || i < 0 || i >= FAR_SIDE_STREAMS) return FAR_EINVAL;
    if (hipEventRecord(d->join[i], d->s[i]) != hipSuccess || hipStreamWaitEvent(main, d->join[i], 0) != hipSuccess) return far_check_launch();
    return FAR_OK;
}
