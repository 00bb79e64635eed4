// Error plumbing and version of the C ABI (include/chipmunk_hip.h).
#include "common.h"

static thread_local char g_last_error[512] = "";

void chipmunk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" const char *chipmunk_last_error(void) { return g_last_error; }
extern "C" int chipmunk_abi_version(void) { return 1; }
