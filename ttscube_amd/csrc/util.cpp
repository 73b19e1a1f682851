// Error state, version and device probing for libttscube_hip.so.
#include "common.hpp"

namespace ttsc {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ttsc

extern "C" const char* ttsc_version(void) { return "ttscube_hip 0.1.0 (gfx950)"; }
extern "C" const char* ttsc_last_error(void) { return ttsc::g_err; }
extern "C" int ttsc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        ttsc::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return n;
}
