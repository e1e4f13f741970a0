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
