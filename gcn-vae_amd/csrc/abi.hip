// Version / error-string part of the C ABI.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace gv {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gv

extern "C" int gv_version(void) { return 100; }
extern "C" const char* gv_last_error_string(void) { return gv::g_err; }
