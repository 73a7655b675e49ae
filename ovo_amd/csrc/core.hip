// core.hip -- error reporting and ABI version of libovo_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ovo_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
const char *ovo_hip_last_error(void) { return g_err; }
int ovo_hip_abi_version(void) { return 1; }
}
