// common.hip -- error text, version string, launch checking.
#include "common.h"
#include <cstring>

namespace votenet {
static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(VOTENET_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return VOTENET_OK;
}
} // namespace votenet

extern "C" const char *votenet_last_error(void) { return votenet::g_err; }
extern "C" const char *votenet_version(void) { return "votenet_hip 0.1 gfx950"; }
