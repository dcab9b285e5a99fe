// anyorder_probe2.hip -- semantics of hipExtAnyOrderLaunch on gfx950, second look:
//  A. timing: K_long's workgroups spin for DIFFERENT times (20 + 5 * (b % 8) us); when do K_next's workgroups start -- one by one as
//     K_long's leave, or after the last one?  (256-thread and full-CU 1024-thread / 64 KB LDS forms)
//  B. visibility: K_w writes a buffer with plain stores, K_r (other XCD: block b reads block b + 1's region) reads it with plain /
//     agent-scope loads: how many stale values when K_r is an any-order launch?
//   hipcc --offload-arch=gfx950 -O3 -o anyorder_probe2 anyorder_probe2.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LDS>
__global__ void k_long(long long *st, int base, int step)
{
    __shared__ char pad[LDS > 0 ? LDS : 4];
    if (LDS > 0 && threadIdx.x == 0) ((volatile char *)pad)[blockIdx.x % LDS] = 1;
    const long long t0 = wall_clock64();
    const int ticks = base + step * (int)(blockIdx.x % 8);
    if (threadIdx.x == 0) st[blockIdx.x * 2] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) st[blockIdx.x * 2 + 1] = wall_clock64();
}
template <int LDS>
__global__ void k_next(long long *st)
{
    __shared__ char pad[LDS > 0 ? LDS : 4];
    if (LDS > 0 && threadIdx.x == 0) ((volatile char *)pad)[blockIdx.x % LDS] = 1;
    if (threadIdx.x == 0) st[blockIdx.x] = wall_clock64();
}

__global__ void k_w(int *buf, int per, int val, int coherent)
{
    int *p = buf + (size_t)blockIdx.x * per;
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        if (coherent) __hip_atomic_store(p + i, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p[i] = val;
    }
}
__global__ void k_r(const int *buf, int per, int val, int coherent, unsigned *stale)
{
    const int *p = buf + (size_t)((blockIdx.x + 1) % gridDim.x) * per;
    unsigned bad = 0;
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        const int v = coherent ? __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[i];
        bad += v != val;
    }
    if (bad) atomicAdd(stale, bad);
}

template <int LDS>
static int timing(const char *what, int threads, int flags_long, int flags_next, hipStream_t s, long long *da, long long *db, int n)
{
    std::vector<long long> a(2 * n), b(n);
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemsetAsync(da, 0, n * 16, s)); CK(hipMemsetAsync(db, 0, n * 8, s)); CK(hipStreamSynchronize(s));
        hipExtLaunchKernelGGL(k_long<LDS>, dim3(n), dim3(threads), 0, s, nullptr, nullptr, flags_long, da, 2000, 500);
        hipExtLaunchKernelGGL(k_next<LDS>, dim3(n), dim3(threads), 0, s, nullptr, nullptr, flags_next, db);
        CK(hipGetLastError()); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(a.data(), da, n * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost));
        long long l0 = a[0];
        for (int i = 0; i < n; i++) l0 = std::min(l0, a[2 * i]);
        std::vector<double> ends, starts;
        for (int i = 0; i < n; i++) { ends.push_back((a[2 * i + 1] - l0) / 100.0); starts.push_back((b[i] - l0) / 100.0); }
        std::sort(ends.begin(), ends.end()); std::sort(starts.begin(), starts.end());
        printf("%-58s long ends: min %.2f q25 %.2f med %.2f q75 %.2f max %.2f | next starts: min %.2f q25 %.2f med %.2f q75 %.2f max %.2f\n", what,
               ends[0], ends[n / 4], ends[n / 2], ends[3 * n / 4], ends[n - 1], starts[0], starts[n / 4], starts[n / 2], starts[3 * n / 4], starts[n - 1]);
    }
    return 0;
}

int main()
{
    const int n = 256;
    long long *da, *db;
    CK(hipMalloc(&da, n * 16)); CK(hipMalloc(&db, n * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    if (timing<0>("A 256 thr: plain, plain", 256, 0, 0, s, da, db, n)) return 1;
    if (timing<0>("A 256 thr: plain, any-order", 256, 0, 1, s, da, db, n)) return 1;
    if (timing<0>("A 256 thr: any-order, any-order", 256, 1, 1, s, da, db, n)) return 1;
    if (timing<65536>("A 1024 thr + 64 KB LDS: plain, plain", 1024, 0, 0, s, da, db, n)) return 1;
    if (timing<65536>("A 1024 thr + 64 KB LDS: plain, any-order", 1024, 0, 1, s, da, db, n)) return 1;
    if (timing<65536>("A 1024 thr + 64 KB LDS: any-order, any-order", 1024, 1, 1, s, da, db, n)) return 1;
    // B. visibility
    const int per = 4096;
    int *buf; unsigned *stale;
    CK(hipMalloc(&buf, (size_t)n * per * 4)); CK(hipMalloc(&stale, 4));
    CK(hipMemset(buf, 0, (size_t)n * per * 4));
    for (int cfg = 0; cfg < 6; cfg++) {
        const int fw = (cfg == 0) ? 0 : 1, fr = (cfg == 0) ? 0 : 1;
        const int cw = (cfg == 2 || cfg == 4 || cfg == 5) ? 1 : 0, cr = (cfg == 3 || cfg == 4 || cfg == 5) ? 1 : 0;
        const int fw2 = cfg == 5 ? 0 : fw;          // cfg 5: writer plain launch, reader any-order, both coherent
        CK(hipMemset(stale, 0, 4));
        for (int it = 1; it <= 300; it++) {
            hipExtLaunchKernelGGL(k_w, dim3(n), dim3(256), 0, s, nullptr, nullptr, fw2, buf, per, it, cw);
            hipExtLaunchKernelGGL(k_r, dim3(n), dim3(256), 0, s, nullptr, nullptr, fr, (const int *)buf, per, it, cr, stale);
        }
        CK(hipStreamSynchronize(s));
        unsigned h = 0; CK(hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost));
        printf("B writer %s launch / %s stores, reader %s launch / %s loads: %u stale values of %d\n", fw2 ? "any-order" : "plain", cw ? "agent-scope" : "plain",
               fr ? "any-order" : "plain", cr ? "agent-scope" : "plain", h, 300 * n * per);
    }
    return 0;
}
