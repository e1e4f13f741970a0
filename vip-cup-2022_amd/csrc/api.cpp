// ABI bookkeeping for libvipcup_hip.so: version and per-thread error text.
#include <stdarg.h>
#include <stdio.h>
#include "vipcup_hip.h"

static thread_local char g_err[512] = "";

void vip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vip_version(void) { return 1000; }
extern "C" const char* vip_last_error(void) { return g_err; }

// Experiments (see common.hpp): 1 when the library was built with VIP_BUILD_EXPERIMENTS=1.  Without them the experimental entry
// points still exist (the ABI of include/vipcup_hip.h does not change with a build flag) and report "not handled here".
#ifndef VIP_BUILD_EXPERIMENTS
#define VIP_BUILD_EXPERIMENTS 0
#endif
extern "C" int vip_experiments_built(void) { return VIP_BUILD_EXPERIMENTS; }
#if !VIP_BUILD_EXPERIMENTS
extern "C" int vip_mbconv_expand_dw_supported(int, int, int, int) { return 0; }
extern "C" int vip_mbconv_expand_dw_f16(const void*, const void*, const void*, const float*, const float*, const float*, void*, int, int, int,
                                        int, int, int, int, int, int, int, int, int, int, int, void*) {
    vip_set_error("vip_mbconv_expand_dw_f16: experimental kernel, not in this build (VIP_BUILD_EXPERIMENTS=1 python build.py)");
    return VIP_ERR_UNSUPPORTED;
}
#endif
