// anyorder_probe.hip -- does a kernel dispatched with hipExtAnyOrderLaunch (AQL packet without the barrier bit) start while the
// previous kernel of the SAME stream is still running on gfx950?  K_long spins ~40 us per workgroup; K_next stamps its start.
//   hipcc --offload-arch=gfx950 -O3 -o anyorder_probe anyorder_probe.hip && ./anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_long(long long *st, int ticks)
{
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) st[blockIdx.x * 2] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) st[blockIdx.x * 2 + 1] = wall_clock64();
}
__global__ void k_next(long long *st)
{
    if (threadIdx.x == 0) st[blockIdx.x] = wall_clock64();
}

static void report(const char *what, const std::vector<long long> &a, const std::vector<long long> &b, int n)
{
    long long l0 = a[0], l1 = a[1], n0 = b[0], n1 = b[0];
    for (int i = 0; i < n; i++) { l0 = std::min(l0, a[2 * i]); l1 = std::max(l1, a[2 * i + 1]); n0 = std::min(n0, b[i]); n1 = std::max(n1, b[i]); }
    printf("%-44s long [0, %.2f] us, next starts %.2f .. %.2f us -> %s\n", what, (l1 - l0) / 100.0, (n0 - l0) / 100.0, (n1 - l0) / 100.0,
           n0 < l1 ? "OVERLAPS the running kernel" : "after the end (gap to first start shown)");
}

int main()
{
    const int n = 256;
    long long *da, *db;
    CK(hipMalloc(&da, n * 16)); CK(hipMalloc(&db, n * 8));
    std::vector<long long> a(2 * n), b(n);
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemsetAsync(da, 0, n * 16, s)); CK(hipMemsetAsync(db, 0, n * 8, s)); CK(hipStreamSynchronize(s));
            const char *what = "";
            if (mode == 0) {
                what = "same stream, plain launches";
                k_long<<<n, 256, 0, s>>>(da, 4000); k_next<<<n, 256, 0, s>>>(db);
            } else if (mode == 1) {
                what = "same stream, next = hipExtAnyOrderLaunch";
                k_long<<<n, 256, 0, s>>>(da, 4000);
                hipExtLaunchKernelGGL(k_next, dim3(n), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, db);
            } else if (mode == 2) {
                what = "same stream, both hipExtAnyOrderLaunch";
                hipExtLaunchKernelGGL(k_long, dim3(n), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, da, 4000);
                hipExtLaunchKernelGGL(k_next, dim3(n), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, db);
            } else if (mode == 3) {
                what = "two streams";
                k_long<<<n, 256, 0, s>>>(da, 4000); k_next<<<n, 256, 0, s2>>>(db);
                CK(hipStreamSynchronize(s2));
            } else {
                what = "graph: captured any-order launches";
                hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                hipExtLaunchKernelGGL(k_long, dim3(n), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, da, 4000);
                hipExtLaunchKernelGGL(k_next, dim3(n), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, db);
                hipError_t e = hipStreamEndCapture(s, &g);
                if (e != hipSuccess || !g) { printf("graph capture of hipExtLaunchKernel: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); break; }
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
            }
            CK(hipGetLastError());
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(a.data(), da, n * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost));
            report(what, a, b, n);
        }
    }
    return 0;
}
