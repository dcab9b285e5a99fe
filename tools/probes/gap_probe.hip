// gap_probe.hip -- what a dependent kernel boundary costs on this box as a function of launch shape:
// workgroup size, workgroups per launch, kernel-argument bytes, dynamic LDS bytes.  200 trivial launches captured in one
// hipGraph, replayed; prints microseconds per launch.  hipcc --offload-arch=gfx950 -O3 -o gap_probe gap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N> struct Args { int *out; int pad[N]; };

template <int TH, int N>
__global__ void __launch_bounds__(TH) k_touch(const Args<N> a)
{
    extern __shared__ char smem[];
    // read the LAST word of the argument block (forces the whole block to be fetched), one tiny store
    if (threadIdx.x == 0) { int v = a.pad[N - 1]; if (smem != nullptr && v == 12345) a.out[blockIdx.x] = v; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int TH, int N>
int run(const char *name, int wgs, size_t lds, int *out, hipStream_t s)
{
    Args<N> a; a.out = out; for (int i = 0; i < N; i++) a.pad[i] = i;
    auto kern = k_touch<TH, N>;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int L = 200;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < L; i++) kern<<<dim3(wgs), dim3(TH), lds, s>>>(a);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    const int R = 10;
    for (int r = 0; r < R; r++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s threads %4d wgs %5d kernarg %4zu B lds %6zu B : %.3f us/launch\n", name, TH, wgs, sizeof(Args<N>), lds, ms * 1000.0f / (R * L));
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return 0;
}

int main()
{
    int *out; CK(hipMalloc(&out, 1 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    run<256, 14>("base", 256, 0, out, s);
    run<512, 14>("512 threads", 256, 0, out, s);
    run<1024, 14>("1024 threads", 256, 0, out, s);
    run<256, 14>("1024 wgs of 256", 1024, 0, out, s);
    run<256, 14>("2048 wgs of 256", 2048, 0, out, s);
    run<512, 14>("512 wgs of 512", 512, 0, out, s);
    run<512, 62>("kernarg 256 B", 256, 0, out, s);
    run<512, 64>("kernarg 264 B", 256, 0, out, s);
    run<512, 126>("kernarg 512 B", 256, 0, out, s);
    run<1024, 62>("1024 thr, kernarg 256 B", 256, 0, out, s);
    run<1024, 64>("1024 thr, kernarg 264 B", 256, 0, out, s);
    run<512, 14>("lds 16 KB", 256, 16 * 1024, out, s);
    run<512, 14>("lds 64 KB", 256, 64 * 1024, out, s);
    run<512, 14>("lds 128 KB", 256, 128 * 1024, out, s);
    run<1024, 14>("1024 thr, lds 16 KB", 256, 16 * 1024, out, s);
    run<64, 14>("64 threads, 1 wg", 1, 0, out, s);
    run<256, 14>("32 wgs of 256 (attention shape)", 32, 0, out, s);
    return 0;
}
