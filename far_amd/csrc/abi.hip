// ABI bookkeeping for libfar_hip.so.
#include "common.h"
extern "C" int far_abi_version(void) { return 1; }
