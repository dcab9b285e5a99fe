// ifa_runtime.hip -- library, device-memory and stream entry points
// (counterparts of CudaUtil, src/common/cuda_util.h:40-61) and the element
// type registry (src/tensor/tensor_common.cc:6-234).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <strings.h>
#include "ifa_host.h"
#include "ifa_device.h"

static thread_local char g_err[512] = "";

int ifa_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" {

const char *ifa_version(void) { return "inferflow_amd 0.1 (gfx950)"; }
const char *ifa_last_error(void) { return g_err; }

int ifa_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int ifa_set_device(int device) { IFA_HIP_CHECK(hipSetDevice(device)); return IFA_OK; }

int ifa_malloc(void **dptr, size_t bytes)
{
    IFA_REQUIRE(dptr != nullptr, "ifa_malloc: null out pointer");
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return ifa_fail(e == hipErrorOutOfMemory ? IFA_ERR_NOMEM : IFA_ERR_HIP,
                                         "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return IFA_OK;
}

int ifa_free(void *dptr) { if (dptr) IFA_HIP_CHECK(hipFree(dptr)); return IFA_OK; }

int ifa_memcpy_h2d(void *dst, const void *src, size_t bytes, ifa_stream s)
{
    IFA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ifa_s(s)));
    return IFA_OK;
}
int ifa_memcpy_d2h(void *dst, const void *src, size_t bytes, ifa_stream s)
{
    IFA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ifa_s(s)));
    return IFA_OK;
}
int ifa_memcpy_d2d(void *dst, const void *src, size_t bytes, ifa_stream s)
{
    IFA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ifa_s(s)));
    return IFA_OK;
}
int ifa_memset(void *dst, int value, size_t bytes, ifa_stream s)
{
    IFA_HIP_CHECK(hipMemsetAsync(dst, value, bytes, ifa_s(s)));
    return IFA_OK;
}
int ifa_stream_create(ifa_stream *out)
{
    IFA_REQUIRE(out != nullptr, "ifa_stream_create: null out pointer");
    hipStream_t s;
    IFA_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = s;
    return IFA_OK;
}
int ifa_stream_destroy(ifa_stream s) { IFA_HIP_CHECK(hipStreamDestroy(ifa_s(s))); return IFA_OK; }
int ifa_stream_sync(ifa_stream s) { IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(s))); return IFA_OK; }

int ifa_block_capacity(int dtype) { return ifa::block_capacity(dtype); }
int ifa_block_bytes(int dtype) { return ifa::block_bytes(dtype); }
size_t ifa_row_bytes(int dtype, size_t cols)
{
    int c = ifa::block_capacity(dtype);
    if (c <= 0) return 0;
    return (cols + (size_t)c - 1) / (size_t)c * (size_t)ifa::block_bytes(dtype);
}

int ifa_dtype_from_name(const char *name)
{   // TensorCommon::InitElementTypeMap, src/tensor/tensor_common.cc:171-205
    static const struct { const char *n; int dt; } tbl[] = {
        {"fp32", IFA_F32}, {"f32", IFA_F32}, {"fp16", IFA_F16}, {"f16", IFA_F16},
        {"q8", IFA_Q8_B32T2}, {"q6", IFA_Q6_B64T1}, {"q5", IFA_Q5_B64T1}, {"q4", IFA_Q4_B32T1A},
        {"q3h", IFA_Q3H_B64T1}, {"q3", IFA_Q3_B32T1B}, {"q2", IFA_Q2_B32T1B},
        {"q8_b32t1", IFA_Q8_B32T1}, {"q8_b32t2", IFA_Q8_B32T2}, {"q6_b64t1", IFA_Q6_B64T1},
        {"q5_b32t1", IFA_Q5_B32T1}, {"q5_b64t1", IFA_Q5_B64T1}, {"q4_b16", IFA_Q4_B16},
        {"q4_b32t1a", IFA_Q4_B32T1A}, {"q4_b32t1b", IFA_Q4_B32T1B}, {"q4_b32t1", IFA_Q4_B32T1A},
        {"q4_b64t1", IFA_Q4_B64T1}, {"q3h_b64t1", IFA_Q3H_B64T1}, {"q3_b32t1a", IFA_Q3_B32T1A},
        {"q3_b32t1b", IFA_Q3_B32T1B}, {"q3_b32t1", IFA_Q3_B32T1B}, {"q2_b32t1a", IFA_Q2_B32T1A},
        {"q2_b32t1b", IFA_Q2_B32T1B}, {"q2_b32t1", IFA_Q2_B32T1B},
    };
    if (!name) return -1;
    for (const auto &e : tbl)
        if (strcasecmp(e.n, name) == 0) return e.dt;
    return -1;
}

} // extern "C"
