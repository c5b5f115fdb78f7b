// Library-wide pieces of the C ABI: version and the per-thread error string.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/unimatch_hip.h"

static thread_local char g_err[512] = "";

void um_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int um_version(void) { return UM_VERSION; }

extern "C" const char* um_last_error_string(void) { return g_err; }
