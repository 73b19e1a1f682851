// Shared helpers for libttscube_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ttscube_hip.h"

namespace ttsc {

void set_error(const char* fmt, ...);

#define TTSC_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ttsc::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return TTSC_EHIP;                                                             \
        }                                                                                 \
    } while (0)

#define TTSC_REQUIRE(cond, ...)             \
    do {                                    \
        if (!(cond)) {                      \
            ttsc::set_error(__VA_ARGS__);   \
            return TTSC_EINVAL;             \
        }                                   \
    } while (0)

// ---- per-device / per-stream state of the multi-workgroup recurrences and of the launch helpers (util.cpp) -------------------
// Number of compute units of the CURRENT device (cached per device id).
int device_cus();
// Raises hipFuncAttributeMaxDynamicSharedMemorySize of `fn` on the CURRENT device to the full 160 KiB once per (device, function).
// Kernels that declare more than 64 KiB of dynamic LDS call this before every launch (a map lookup under a mutex).
int ensure_full_lds(const void* fn);
// Hand-off area of a split recurrence: `nwords` counters, the launch's abort word (both re-zeroed on the stream before every
// launch: an aborted launch cannot poison the next one) and a STICKY copy of the abort that survives until handoff_status()
// has reported it, plus an optional exchange buffer.  One area per (device, stream, tag): two handles driven on two streams of one device — or on two devices of
// one process — never share counters, rings or abort words.  The buffer only grows (re-allocation synchronises the device).
struct HandoffArea {
    unsigned* words = nullptr;
    size_t nwords = 0;
    void* buf = nullptr;
    size_t buf_bytes = 0;
    unsigned* abort_word() const { return words + nwords; }   // [0] this launch, [1] sticky
    hipError_t rearm(hipStream_t s) const { return hipMemsetAsync(words, 0, (nwords + 1) * sizeof(unsigned), s); }
};
HandoffArea* handoff_area(const char* tag, hipStream_t stream, size_t nwords, size_t buf_bytes);
// OR of the abort words of every area of `tag` on the CURRENT device, each cleared once reported; synchronises the device.
// 0 = every hand-off completed, 1 = a bounded spin timed out, -1 = HIP error.
int handoff_status(const char* tag);
// the same for the areas of every tag that belong to `stream`, waiting for that stream only; bit mask lstm 1 | gru 2 | melar 4
int handoff_status_stream(hipStream_t stream);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// floor division for possibly negative numerators
inline int64_t floor_div(int64_t a, int64_t b) {
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

}  // namespace ttsc
