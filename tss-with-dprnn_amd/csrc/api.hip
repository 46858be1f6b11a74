// Error plumbing and ABI bookkeeping of libdprnn_mi355x.so (see include/dprnn_mi355x.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void dprnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int dprnn_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dprnn_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

extern "C" const char* dprnn_last_error(void) { return g_err; }
extern "C" int dprnn_abi_version(void) { return 1; }
