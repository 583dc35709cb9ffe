// Library-level entry points: version, thread-local error string.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void swv2_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int swv2_version(void) { return SWV2_VERSION; }
extern "C" const char* swv2_last_error(void) { return g_err; }
