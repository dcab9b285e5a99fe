// Probe: does overlapping consecutive weight-streaming kernels (two streams, data dependence carried by
// a device flag instead of the kernel boundary) beat the serial chain on MI355X?
// Build: hipcc -O3 --offload-arch=gfx950 -o overlap_probe overlap_probe.hip ; run: ./overlap_probe
// Every spin is bounded (gives up and raises err) so the probe cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;
constexpr int XN = 4096;          // activation elements (half → 8 KB)
constexpr int NB = 4;             // u32x4 per thread per batch (NB*16*256 = 16 KiB per WG batch)

template <bool OVL>
__global__ __launch_bounds__(TPB) void k_probe(const u32x4* __restrict__ W, int batches, const unsigned* x, unsigned* y,
                                               unsigned* done_prev, unsigned target, unsigned* done_me, unsigned* err, unsigned long long* tl) {
  __shared__ unsigned xs[XN / 2];
  const int t = threadIdx.x;
  unsigned long long ts0 = wall_clock64(), ts1 = ts0; int spins_rec = 0;
  const u32x4* w = W + (size_t)blockIdx.x * batches * NB * TPB + t;
  u32x4 a[NB], b[NB];
#pragma unroll
  for (int i = 0; i < NB; i++) a[i] = __builtin_nontemporal_load(w + i * TPB);
  if (OVL) {
    if (t < 64) {
      int spins = 0;
      for (;;) {
        bool ok = true;
        for (unsigned i = t; i < target; i += 64) ok &= __hip_atomic_load(done_prev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(4);
        if (++spins > 20000 || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(err, 1u); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      spins_rec = spins; ts1 = wall_clock64();
    }
    __syncthreads();
  }
  for (int i = t; i < XN / 2; i += TPB) xs[i] = __hip_atomic_load(x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int acc = 0;
  auto eat = [&](const u32x4& v, int k) {
    acc = __builtin_amdgcn_sdot4((int)v.x, (int)xs[(k * 4 + 0 + t) & (XN / 2 - 1)], acc, false);
    acc = __builtin_amdgcn_sdot4((int)v.y, (int)xs[(k * 4 + 1 + t) & (XN / 2 - 1)], acc, false);
    acc = __builtin_amdgcn_sdot4((int)v.z, (int)xs[(k * 4 + 2 + t) & (XN / 2 - 1)], acc, false);
    acc = __builtin_amdgcn_sdot4((int)v.w, (int)xs[(k * 4 + 3 + t) & (XN / 2 - 1)], acc, false);
  };
  for (int bt = 0; bt < batches; bt += 2) {
    const u32x4* wn = w + (size_t)(bt + 1 < batches ? bt + 1 : bt) * NB * TPB;
#pragma unroll
    for (int i = 0; i < NB; i++) b[i] = __builtin_nontemporal_load(wn + i * TPB);
#pragma unroll
    for (int i = 0; i < NB; i++) eat(a[i], bt * NB + i);
    const u32x4* wn2 = w + (size_t)(bt + 2 < batches ? bt + 2 : bt) * NB * TPB;
#pragma unroll
    for (int i = 0; i < NB; i++) a[i] = __builtin_nontemporal_load(wn2 + i * TPB);
    if (bt + 1 < batches) {
#pragma unroll
      for (int i = 0; i < NB; i++) eat(b[i], (bt + 1) * NB + i);
    }
  }
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if ((t & 63) == 0) {
    unsigned slot = (blockIdx.x * 4 + (t >> 6)) & (XN / 2 - 1);
    __hip_atomic_store(y + slot, (unsigned)acc | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (t == 0) __hip_atomic_store(done_me + blockIdx.x, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (t == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
    unsigned long long* o = tl + (blockIdx.x ? 4 : 0);
    o[0] = ts0; o[1] = ts1; o[2] = wall_clock64(); o[3] = spins_rec;
  }
}

struct Step { size_t off16; int batches; int grid; };

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 32;
  const int grid = argc > 2 ? atoi(argv[2]) : 512;
  const double mb[4] = {31.46, 10.49, 56.36, 28.18};
  const int distinct = 3;  // distinct layers of weights (380 MB > MALL)
  std::vector<Step> proto;
  size_t off = 0;
  for (int l = 0; l < distinct; l++)
    for (int k = 0; k < 4; k++) {
      int batches = (int)(mb[k] * 1e6 / ((double)grid * NB * TPB * 16) + 0.5);
      if (batches < 1) batches = 1;
      proto.push_back({off, batches, grid});
      off += (size_t)grid * batches * NB * TPB;
    }
  u32x4* W; CK(hipMalloc(&W, off * 16)); CK(hipMemset(W, 0x11, off * 16));
  unsigned *x0, *x1, *done, *err;
  CK(hipMalloc(&x0, XN * 2)); CK(hipMalloc(&x1, XN * 2)); CK(hipMemset(x0, 1, XN * 2)); CK(hipMemset(x1, 1, XN * 2));
  const int nk = layers * 4;
  CK(hipMalloc(&done, (size_t)(nk + 1) * grid * 4)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  unsigned long long* tl; CK(hipMalloc(&tl, nk * 64)); CK(hipMemset(tl, 0, nk * 64));
  hipStream_t s[2]; CK(hipStreamCreate(&s[0])); CK(hipStreamCreate(&s[1]));
  hipEvent_t fork, join, t0, t1; CK(hipEventCreate(&fork)); CK(hipEventCreate(&join)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  double bytes = 0;
  for (int i = 0; i < nk; i++) { const Step& p = proto[i % proto.size()]; bytes += (double)p.grid * p.batches * NB * TPB * 16; }

  hipGraphExec_t exec[2];
  auto enqueue = [&](int mode) {
    CK(hipMemsetAsync(done, 0, (size_t)(nk + 1) * grid * 4, s[0]));
    if (mode == 1) { CK(hipEventRecord(fork, s[0])); CK(hipStreamWaitEvent(s[1], fork, 0)); }
    for (int i = 0; i < nk; i++) {
      const Step& p = proto[i % proto.size()];
      hipStream_t st = mode == 1 ? s[i & 1] : s[0];
      unsigned* xi = (i & 1) ? x1 : x0; unsigned* yo = (i & 1) ? x0 : x1;
      if (mode == 1 && i > 0)
        hipLaunchKernelGGL(k_probe<true>, dim3(p.grid), dim3(TPB), 0, st, W + p.off16, p.batches, xi, yo, done + (size_t)(i - 1) * grid, (unsigned)proto[(i - 1) % proto.size()].grid, done + (size_t)i * grid, err, tl + 8 * i);
      else
        hipLaunchKernelGGL(k_probe<false>, dim3(p.grid), dim3(TPB), 0, st, W + p.off16, p.batches, xi, yo, done, 0u, done + (size_t)i * grid, err, tl + 8 * i);
    }
    if (mode == 1) { CK(hipEventRecord(join, s[1])); CK(hipStreamWaitEvent(s[0], join, 0)); }
  };
  for (int mode = 0; mode < 2; mode++) {
    hipGraph_t g;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
    enqueue(mode);
    CK(hipStreamEndCapture(s[0], &g));
    CK(hipGraphInstantiate(&exec[mode], g, nullptr, nullptr, 0));
  }
  std::vector<unsigned> ref(XN / 2), got(XN / 2);
  for (int mode = 0; mode < 4; mode++) {
    auto go = [&]() { if (mode < 2) CK(hipGraphLaunch(exec[mode], s[0])); else enqueue(mode == 2 ? 1 : 0); };
    CK(hipMemset(err, 0, 4));
    for (int w = 0; w < 3; w++) go();
    CK(hipStreamSynchronize(s[0]));
    const int reps = 20;
    CK(hipEventRecord(t0, s[0]));
    for (int r = 0; r < reps; r++) go();
    CK(hipEventRecord(t1, s[0])); CK(hipStreamSynchronize(s[0]));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(mode ? got.data() : ref.data(), (nk & 1) ? x0 : x1, XN * 2, hipMemcpyDeviceToHost));
    printf("%s: %d kernels grid %d  %.1f us/chain  %.2f us/kernel  %.2f TB/s  err=%u\n", mode == 0 ? "serial graph " : mode == 1 ? "overlap graph" : mode == 2 ? "overlap direct" : "serial direct", nk, grid,
           ms * 1e3 / reps, ms * 1e3 / reps / nk, bytes * reps / (ms * 1e-3) / 1e12, e); fflush(stdout);
    { std::vector<unsigned long long> h(nk * 8); CK(hipMemcpy(h.data(), tl, nk * 64, hipMemcpyDeviceToHost));
      int n = nk < 10 ? nk : 10; unsigned long long z = h[0];
      for (int i = 0; i < n; i++) printf("   k%d wg0: start %.2f flag %.2f end %.2f spins %llu | wgLast: start %.2f flag %.2f end %.2f\n", i,
        (double)(long long)(h[i*8]-z)/100.0, (double)(long long)(h[i*8+1]-z)/100.0, (double)(long long)(h[i*8+2]-z)/100.0, h[i*8+3],
        (double)(long long)(h[i*8+4]-z)/100.0, (double)(long long)(h[i*8+5]-z)/100.0, (double)(long long)(h[i*8+6]-z)/100.0); }
  }
  setvbuf(stdout, nullptr, _IONBF, 0);
  int bad = 0; for (int i = 0; i < XN / 2; i++) bad += ref[i] != got[i];
  printf("mismatching outputs: %d\n", bad);
  return 0;
}
