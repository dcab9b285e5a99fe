// ifa_runtime.hip -- library, device-memory and stream entry points
// (counterparts of CudaUtil, src/common/cuda_util.h:40-61) and the element
// type registry (src/tensor/tensor_common.cc:6-234).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <strings.h>
#include <algorithm>
#include <string>
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

// ------------------------------------------------------------------ in-launch waits (ifa_host.h)
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>

namespace ifa {

static std::mutex g_wait_mu;
static std::map<int, unsigned *> g_wait_words;
static std::atomic<int> g_waits_off{-1};        // -1: not decided yet (environment), 0 on, 1 off

unsigned *wait_err_word()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(g_wait_mu);
    auto it = g_wait_words.find(dev);
    if (it != g_wait_words.end() && it->second) return it->second;
    // (a failed allocation -- e.g. the first use fell inside a stream capture -- is NOT remembered: the next call tries again, and
    //  the launchers below take their non-waiting kernels while the word is missing)
    unsigned *p = nullptr;
    if (hipHostMalloc((void **)&p, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    memset(p, 0, 64);
    g_wait_words[dev] = p;
    return p;
}

bool waits_enabled()
{
    int v = g_waits_off.load(std::memory_order_relaxed);
    if (v < 0) { v = getenv("IFA_NO_INLAUNCH_WAITS") ? 1 : 0; g_waits_off.store(v, std::memory_order_relaxed); }
    return v == 0;
}

void waits_disable(const char *why)
{
    if (g_waits_off.exchange(1) != 1)
        fprintf(stderr, "inferflow_amd: launches that wait for sibling workgroups are switched off for this process (%s); the non-waiting kernels serve from here on\n", why);
}

int wait_err_check(const char *who)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return IFA_OK; }
    unsigned *p = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_wait_mu);
        auto it = g_wait_words.find(dev);
        if (it == g_wait_words.end() || !it->second) return IFA_OK;
        p = it->second;
    }
    const unsigned code = *(volatile unsigned *)p;
    if (code == 0) return IFA_OK;
    *(volatile unsigned *)p = 0;
    waits_disable("a bounded wait inside a launch timed out: the partner workgroups were not resident -- another process, a CU mask or a side stream holds part of the device");
    return ifa_fail(IFA_ERR_STATE, "%s: a wait for a sibling workgroup inside a launch timed out (code 0x%x: 0x7_ rows GEMM K parts, 0x8_ split-K GEMM); "
                    "the results of this call are not valid -- repeat it: the waiting launches are off now", who, code);
}

// ROC_GLOBAL_CU_MASK / HSA_CU_MASK name the CUs a process may use; the runtime keeps reporting the device's full count.
// Forms: a hex mask ("0xffff", ROC_GLOBAL_CU_MASK) or "<gpu>:<ranges>" lists separated by ';' ("0:0-31;1:0-15", HSA_CU_MASK).
// Returns the CU count the mask leaves for `device`, or device_cus when the text names no mask for it / cannot be read.
int visible_cus_uncached();
static int cus_from_mask(const char *mask, int device, int device_cus)
{
    if (!mask || !*mask) return device_cus;
    std::string m(mask);
    if (m.find(':') == std::string::npos) {          // plain hex mask
        size_t i = (m.size() > 2 && m[0] == '0' && (m[1] == 'x' || m[1] == 'X')) ? 2 : 0;
        int bits = 0; bool any = false;
        for (; i < m.size(); i++) {
            const char ch = m[i];
            int v = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : (ch >= 'A' && ch <= 'F') ? ch - 'A' + 10 : -1;
            if (v < 0) return device_cus;
            any = true;
            bits += __builtin_popcount((unsigned)v);
        }
        return any && bits > 0 ? std::min(bits, device_cus) : device_cus;
    }
    size_t pos = 0;
    while (pos < m.size()) {
        size_t end = m.find(';', pos);
        if (end == std::string::npos) end = m.size();
        const std::string ent = m.substr(pos, end - pos);
        pos = end + 1;
        const size_t colon = ent.find(':');
        if (colon == std::string::npos) continue;
        {   // GPU list in front of ':' -- ids and ranges separated by ',' ("0,2-3:0-31"); an entry that cannot be read counts as "masked, size unknown"
            const std::string gl = ent.substr(0, colon);
            bool match = false, bad = gl.empty();
            size_t gp = 0;
            while (gp < gl.size() && !bad) {
                size_t ge = gl.find(',', gp);
                if (ge == std::string::npos) ge = gl.size();
                const std::string gr = gl.substr(gp, ge - gp);
                gp = ge + 1;
                if (gr.empty() || gr.find_first_not_of("0123456789-") != std::string::npos) { bad = true; break; }
                const size_t gd = gr.find('-');
                const int ga = atoi(gr.c_str()), gb = gd == std::string::npos ? ga : atoi(gr.c_str() + gd + 1);
                if (device >= ga && device <= gb) match = true;
            }
            if (bad) return 0;          // conservative: no launch waits when a mask is set but unreadable
            if (!match) continue;
        }
        int n = 0;
        size_t q = colon + 1;
        while (q < ent.size()) {
            size_t e2 = ent.find(',', q);
            if (e2 == std::string::npos) e2 = ent.size();
            const std::string r = ent.substr(q, e2 - q);
            q = e2 + 1;
            const size_t dash = r.find('-');
            const int a = atoi(r.c_str()), b = dash == std::string::npos ? a : atoi(r.c_str() + dash + 1);
            if (b >= a) n += b - a + 1;
        }
        return n > 0 ? std::min(n, device_cus) : device_cus;
    }
    return device_cus;
}

static std::map<int, int> g_visible_cus;      // per device: the environment is read once
int visible_cus()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    {
        std::lock_guard<std::mutex> lk(g_wait_mu);
        auto it = g_visible_cus.find(dev);
        if (it != g_visible_cus.end()) return it->second;
    }
    const int v = visible_cus_uncached();
    std::lock_guard<std::mutex> lk(g_wait_mu);
    g_visible_cus[dev] = v;
    return v;
}
int visible_cus_uncached()
{
    int dev = 0;
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    else (void)hipGetLastError();
    if (const char *f = getenv("IFA_VISIBLE_CUS")) { const int v = atoi(f); if (v > 0) return std::min(v, n); }     // (tests; an operator who knows better)
    n = cus_from_mask(getenv("ROC_GLOBAL_CU_MASK"), dev, n);
    n = cus_from_mask(getenv("HSA_CU_MASK"), dev, n);
    return n;
}

static std::map<std::pair<const void *, size_t>, int> g_occ;
bool wait_grid_fits(const void *kernel, int threads, size_t smem_bytes, long long grid)
{
    if (!waits_enabled()) return false;
    if (!wait_err_word()) return false;      // a timed-out wait could not be reported: the launches that wait are not chosen
    int occ = 0;
    {
        std::lock_guard<std::mutex> lk(g_wait_mu);
        auto it = g_occ.find(std::make_pair(kernel, smem_bytes * 4096 + (size_t)threads));
        if (it != g_occ.end()) occ = it->second;
        else {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, smem_bytes) != hipSuccess) { (void)hipGetLastError(); occ = 0; }
            g_occ[std::make_pair(kernel, smem_bytes * 4096 + (size_t)threads)] = occ;
        }
    }
    return ifa_wait_grid_decision(occ, visible_cus(), grid) != 0;
}

} // namespace ifa

extern "C" {

// host-only (tests): the co-residency rule and the CU-mask reader behind ifa::wait_grid_fits
int ifa_wait_grid_decision(int blocks_per_cu, int visible_cus, long long grid)
{
    return blocks_per_cu >= 1 && visible_cus >= 1 && grid >= 1 && grid <= (long long)blocks_per_cu * (long long)visible_cus ? 1 : 0;
}
int ifa_visible_cus_from_mask(const char *mask, int device, int device_cus) { return ifa::cus_from_mask(mask, device, device_cus); }
int ifa_inlaunch_waits_enabled(void) { return ifa::waits_enabled() ? 1 : 0; }

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
int ifa_stream_sync(ifa_stream s)
{
    IFA_HIP_CHECK(hipStreamSynchronize(ifa_s(s)));
    return ifa::wait_err_check("ifa_stream_sync");      // (a launch that waited for sibling workgroups in vain left its code: the synchronising call reports it)
}

// (ifa::Q3H_NATIVE is a kernel-format id inside the library, not an element type: the registry does not know it)
int ifa_block_capacity(int dtype) { return dtype == ifa::Q3H_NATIVE ? 0 : ifa::block_capacity(dtype); }
int ifa_block_bytes(int dtype) { return dtype == ifa::Q3H_NATIVE ? 0 : ifa::block_bytes(dtype); }
size_t ifa_row_bytes(int dtype, size_t cols)
{
    int c = ifa_block_capacity(dtype);
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
