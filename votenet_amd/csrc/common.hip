// common.hip -- error text, version string, launch checking.
#include "common.h"
#include <atomic>
#include <cstdlib>
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

static std::atomic<int> g_debug_enabled{-1}; // -1: not decided yet (the environment is read once, at the first switch call)
bool debug_gate(const char *name)
{
    int v = g_debug_enabled.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("VOTENET_DEBUG");
        const int want = (e && e[0] && strcmp(e, "0") != 0) ? 1 : 0;
        int expect = -1;
        g_debug_enabled.compare_exchange_strong(expect, want);
        v = g_debug_enabled.load(std::memory_order_relaxed);
    }
    if (v <= 0) set_error(VOTENET_E_INVALID_ARGUMENT, "%s ignored: debug switches are disabled (call votenet_debug_enable(1) or set VOTENET_DEBUG=1)", name);
    return v > 0;
}
} // namespace votenet

extern "C" void votenet_debug_enable(int on) { votenet::g_debug_enabled.store(on ? 1 : 0); }
extern "C" int votenet_debug_enabled(void) { return votenet::g_debug_enabled.load() > 0 ? 1 : 0; }
extern "C" const char *votenet_last_error(void) { return votenet::g_err; }
extern "C" const char *votenet_version(void) { return "votenet_hip 0.1 gfx950"; }
