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

// ---- launches that wait for sibling workgroups INSIDE the launch (fused QKV + attention, split-K halves of the large-tile GEMM,
// K parts of the rows GEMM): every such wait is bounded; a wait that gives up stores a code in the device's wait-error word and
// carries on (its results are garbage, the call fails); the host looks at the word after the next synchronisation, reports the
// failure and switches the waiting launches OFF for the rest of the process (the non-waiting kernels serve from then on).
// Nothing here kills the HIP context.  Progress of such a launch needs all partners resident at once: wait_grid_fits() asks the
// occupancy calculator and the CUs this process may use (a CU mask or a partitioned device shrinks that number) before one is chosen.
namespace ifa {
constexpr long long WAIT_TIMEOUT_TICKS = 20000000ll;            // 0.2 s of the 100 MHz wall clock: a legitimate wait is microseconds
unsigned *wait_err_word();                                       // device-visible (pinned, mapped) word of the current device; null: unavailable
bool waits_enabled();                                            // false once a wait has timed out in this process (or IFA_NO_INLAUNCH_WAITS is set)
void waits_disable(const char *why);
// after a synchronisation of the stream: IFA_OK, or the failure a timed-out wait left (word cleared, waits switched off)
int wait_err_check(const char *who);
int visible_cus();                                               // CUs of the current device this process can occupy
bool wait_grid_fits(const void *kernel, int threads, size_t smem_bytes, long long grid);
}
