// ABI bookkeeping for libfar_hip.so.
#include "common.h"

static thread_local int g_last_hip_error = 0;
void far_record_hip_error(int e) { g_last_hip_error = e; }

extern "C" int far_abi_version(void) { return 1; }
// hipError_t of the most recent failed launch on this thread (0 = none); for diagnostics after a -5 return.
extern "C" int far_last_hip_error(void) { return g_last_hip_error; }
