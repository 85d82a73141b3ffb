// Error plumbing and version of libvaw_hip.so (the kernels' extern "C" entry points live next to them).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vaw_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vaw_version(void) { return 100; }   // 0.1.0
extern "C" const char* vaw_last_error_string(void) { return g_err; }
