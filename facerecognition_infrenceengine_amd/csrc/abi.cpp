// Error plumbing + version of the C ABI (include/frhip.h).
#include "common.h"

static thread_local char g_err[512] = "";

void fr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int fr_version(void) { return 100; }
extern "C" const char* fr_last_error_string(void) { return g_err; }
extern "C" int fr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
