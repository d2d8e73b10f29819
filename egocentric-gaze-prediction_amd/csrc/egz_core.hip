// libegaze_hip.so -- version / error plumbing.
#include "egz_common.h"
#include <cstdarg>

static thread_local char g_err[512] = "";

void egz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

EGZ_API const char* egz_version(void) { return "egaze-hip 0.1 (gfx950)"; }
EGZ_API const char* egz_last_error(void) { return g_err; }
