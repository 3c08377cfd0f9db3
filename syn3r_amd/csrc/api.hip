// Error string, version and arch queries of the C-ABI (include/syn3r_hip.h).
#include "common.h"
#include <string.h>

namespace syn3r {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace syn3r

extern "C" const char* syn3r_last_error(void) { return syn3r::g_err; }
extern "C" int syn3r_version(void) { return 100; }
extern "C" const char* syn3r_arch(void) { return "gfx950"; }
