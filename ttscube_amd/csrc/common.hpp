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

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// floor division for possibly negative numerators
inline int64_t floor_div(int64_t a, int64_t b) {
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

}  // namespace ttsc
