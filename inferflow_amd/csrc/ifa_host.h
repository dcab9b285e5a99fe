// ifa_host.h -- host-side helpers for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/inferflow_amd.h"

// Records the message for ifa_last_error() and returns `code`.
int ifa_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define IFA_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return ifa_fail(IFA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define IFA_LAUNCH_CHECK() IFA_HIP_CHECK(hipGetLastError())

#define IFA_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return ifa_fail(IFA_ERR_ARG, __VA_ARGS__); \
    } while (0)

static inline hipStream_t ifa_s(ifa_stream s) { return reinterpret_cast<hipStream_t>(s); }
static inline unsigned ifa_cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
