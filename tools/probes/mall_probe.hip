// mall_probe.hip -- round 4: can the 256 MiB Infinity Cache hide a decode layer's weight stream behind the fixed
// costs of its five dependent launches?
//   A. bandwidth of a streaming read whose bytes were read once before (by default-policy or nt loads), by size and by
//      how much other traffic came in between (retention);
//   B. a chain shaped like the decode step (per layer 31.46 / 10.49 / 56.36 / 28.18 MB streaming kernels that each wait
//      for an 8 KB vector of the previous one + a 32-workgroup latency-bound "attention" kernel), replayed as one
//      hipGraph on stream 0, alone and next to a PREFETCHER kernel launched directly on stream 1: one small wave per CU
//      that walks the same weights ahead of the chain (default-policy loads dumped into LDS), throttled by a progress
//      word the chain's kernels bump;
//   C. the same chain with the prefetch done by otherwise idle workgroups of the attention kernel.
// Build: hipcc -O3 --offload-arch=gfx950 -o mall_probe mall_probe.hip ; every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4 *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// ---------------------------------------------------------------- A: plain streaming read, 256 x 1024 threads
// every thread keeps NL 16-byte loads in flight; stamps: [wg][2] start / end
template <bool NT, int NL>
__global__ void __launch_bounds__(1024) k_read(const u32x4 *__restrict__ p, size_t n16, unsigned *sink, unsigned long long *stamps)
{
    const unsigned long long t0 = wall_clock64();
    const size_t stride = (size_t)gridDim.x * 1024;
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    unsigned acc = 0;
    for (; i + (NL - 1) * stride < n16; i += NL * stride) {
        u32x4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) v[k] = ld16<NT>(p + i + k * stride);
#pragma unroll
        for (int k = 0; k < NL; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    for (; i < n16; i += stride) { u32x4 v = ld16<NT>(p + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345679u) sink[0] = acc;
    if (stamps && threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = wall_clock64(); }
}

// ---------------------------------------------------------------- B: the chain
constexpr int XN = 4096;            // the activation: 4096 halves = 2048 dwords = 8 KB
struct MainArgs {
    const u32x4 *W; size_t n16;     // this kernel's weights
    const unsigned *x; unsigned *y; // activation in / out (2048 dwords each)
    unsigned *prog;                 // progress word (bumped by workgroup 0 at the start)
    unsigned long long *stamps;     // [4]: wg0 start, x arrived, end; or null
    // C: prefetch by surplus workgroups (attention kernel only)
    const u32x4 *P; size_t pn16;
    // D: address-translation warm-up: workgroups 0-7 (one per XCD) touch one line per `tstride16` units of the NEXT kernel's weights
    const u32x4 *T; size_t tn16, tstride16;
};

// a GEMV-shaped kernel: x first (dependent on the previous kernel), a prologue of ~PROLOG sleep units, then all of the
// thread's weights in flight at once (<= NL loads of 16 bytes), reduce, one output dword per wave
template <bool NT, int NL>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(112))) k_main(const MainArgs a)
{
    __shared__ unsigned xs[XN / 2];
    const int t = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    if (t == 0 && blockIdx.x == 0 && a.prog) __hip_atomic_fetch_add(a.prog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned xa = a.x[t], xb = a.x[t + 1024];
    unsigned tacc = 0;
    if (a.T && blockIdx.x < 8) {
        for (size_t i = (size_t)t * a.tstride16; i < a.tn16; i += (size_t)1024 * a.tstride16) tacc ^= a.T[i].x;
    }
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 1024;
    const size_t i0 = (size_t)blockIdx.x * 1024 + t;
    u32x4 v[NL];
#pragma unroll
    for (int k = 0; k < 2; k++) { const size_t i = i0 + k * stride; v[k] = ld16<NT>(a.W + (i < a.n16 ? i : i0)); }
    xs[t] = xa; xs[t + 1024] = xb;
    const unsigned long long t1 = wall_clock64();
    __syncthreads();
    // ~1.5 us of "norm + quantise" VALU work
    unsigned q = xs[(t * 7) & (XN / 2 - 1)];
    for (int r = 0; r < 160; r++) q = q * 1664525u + 1013904223u;
    xs[t] ^= q & 1;
    __syncthreads();
#pragma unroll
    for (int k = 2; k < NL; k++) { const size_t i = i0 + k * stride; v[k] = ld16<NT>(a.W + (i < a.n16 ? i : i0)); }
    int acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        acc = __builtin_amdgcn_sdot4((int)v[k].x, (int)xs[(k * 4 + 0 + t) & (XN / 2 - 1)], acc, false);
        acc = __builtin_amdgcn_sdot4((int)v[k].y, (int)xs[(k * 4 + 1 + t) & (XN / 2 - 1)], acc, false);
        acc = __builtin_amdgcn_sdot4((int)v[k].z, (int)xs[(k * 4 + 2 + t) & (XN / 2 - 1)], acc, false);
        acc = __builtin_amdgcn_sdot4((int)v[k].w, (int)xs[(k * 4 + 3 + t) & (XN / 2 - 1)], acc, false);
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (tacc == 0x12345679u) acc ^= 1;
    if ((t & 63) == 0) a.y[(blockIdx.x * 16 + (t >> 6)) & (XN / 2 - 1)] = (unsigned)acc | 1u;
    if (a.stamps && t == 0 && blockIdx.x == 0) { a.stamps[0] = t0; a.stamps[1] = t1; a.stamps[2] = wall_clock64(); }
}

// the "attention" kernel: 32 workgroups do a dependent latency chain (~3 us), the others (C) prefetch P
__global__ void __launch_bounds__(256) k_attn(const MainArgs a, int work_wgs)
{
    const int t = threadIdx.x;
    if ((int)blockIdx.x < work_wgs) {
        const unsigned long long t0 = wall_clock64();
        unsigned q = a.x[(blockIdx.x * 64 + t) & (XN / 2 - 1)];
        // three dependent round trips to memory + some ALU
        for (int r = 0; r < 3; r++) q = a.x[(q + r) & (XN / 2 - 1)] + (q & 1);
        for (int r = 0; r < 200; r++) q = q * 1664525u + 1013904223u;
        if (t < 64) a.y[(blockIdx.x * 64 + t) & (XN / 2 - 1)] = q | 1u;
        if (a.stamps && t == 0 && blockIdx.x == 0) { a.stamps[0] = t0; a.stamps[1] = t0; a.stamps[2] = wall_clock64(); }
        return;
    }
    // C: surplus workgroups read their slice of P with default-policy loads (allocate in L2 / Infinity Cache)
    const int pw = gridDim.x - work_wgs;
    const size_t per = (a.pn16 + pw - 1) / pw;
    const size_t b = (size_t)(blockIdx.x - work_wgs) * per, e = ((b + per) < a.pn16 ? (b + per) : a.pn16);
    unsigned acc = 0;
    for (size_t i = b + t; i < e; i += 256 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const size_t j = i + (size_t)k * 256; v[k] = a.P[j < e ? j : i]; }
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345679u) a.y[0] = acc;
}

// the prefetcher: one workgroup of PT threads per CU; segment s = [seg_off[s], seg_off[s + 1]) in 16-byte units, entered
// once prog >= base + s - ahead (s counts the chain's weight kernels)
struct Seg { size_t off16, n16; int gate; };
template <int PT>
__global__ void __launch_bounds__(PT) __attribute__((amdgpu_num_vgpr(48))) k_prefetch(const u32x4 *__restrict__ W, const Seg *__restrict__ segs, int nseg, const unsigned *prog,
                                                      unsigned base, int ahead, unsigned *err, unsigned *sink, unsigned long long *pst)
{
    const int t = threadIdx.x;
    unsigned acc = 0;
    for (int s = 0; s < nseg; s++) {
        const Seg sg = segs[s];
        const int need = sg.gate - ahead;
        if (need > 0) {
            int spins = 0;
            while ((int)(__hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base) < need) {
                __builtin_amdgcn_s_sleep(32);
                if (++spins > 200000) { if (t == 0) atomicExch(err, 2u); return; }
            }
        }
        if (pst && t == 0 && blockIdx.x == 0 && s < 64) pst[s] = wall_clock64();
        const size_t per = (sg.n16 + gridDim.x - 1) / gridDim.x;
        const size_t b = sg.off16 + (size_t)blockIdx.x * per, e = ((b + per) < (sg.off16 + sg.n16) ? (b + per) : (sg.off16 + sg.n16));
        for (size_t i = b + t; i < e; i += PT * 8) {
            u32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const size_t j = i + (size_t)k * PT; v[k] = W[j < e ? j : i]; }
#pragma unroll
            for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

static double span_us(const std::vector<unsigned long long> &st, int wgs)
{
    unsigned long long lo = ~0ull, hi = 0;
    for (int i = 0; i < wgs; i++) { lo = std::min(lo, st[i * 2]); hi = std::max(hi, st[i * 2 + 1]); }
    return (double)(hi - lo) / 100.0;
}

int main(int argc, char **argv)
{
    const int layers = argc > 1 ? atoi(argv[1]) : 32;
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    unsigned *sink; CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
    unsigned long long *stamps; CK(hipMalloc(&stamps, 256 * 16));
    std::vector<unsigned long long> hst(512);

    // ------------------------------------------------------------ A
    if (!(argc > 2 && atoi(argv[2]) == 4)) {
        const size_t pool_bytes = (size_t)1536 << 20;
        u32x4 *pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0x11, pool_bytes));
        const size_t flush16 = ((size_t)768 << 20) / 16;          // the upper half is the "other traffic"
        const u32x4 *flush = pool + flush16;
        auto rd = [&](bool nt, const u32x4 *p, size_t n16, bool stamp) {
            if (nt) hipLaunchKernelGGL((k_read<true, 8>), dim3(256), dim3(1024), 0, s0, p, n16, sink, stamp ? stamps : nullptr);
            else hipLaunchKernelGGL((k_read<false, 8>), dim3(256), dim3(1024), 0, s0, p, n16, sink, stamp ? stamps : nullptr);
        };
        printf("A. second read of S MB after `between` MB of other reads (first / second policy), in-kernel span of the second read\n");
        const double sizes[] = {10.49, 28.18, 56.36, 126.5, 200.0};
        const int betweens[] = {0, 64, 192};
        for (double mb : sizes) {
            const size_t n16 = (size_t)(mb * 1e6 / 16);
            for (int p1 = 0; p1 < 3; p1++) {                 // 0: cold (no first read), 1: default first, 2: nt first
                for (int p2 = 0; p2 < 2; p2++) {
                    for (int bi = 0; bi < 3; bi++) {
                        if (p1 == 0 && bi > 0) continue;
                        double best = 1e9, sum = 0; const int reps = 5;
                        for (int r = 0; r < reps; r++) {
                            rd(true, flush, flush16, false);                               // flush: 768 MB of other data
                            if (p1) rd(p1 == 2, pool, n16, false);
                            if (betweens[bi]) rd(true, flush, ((size_t)betweens[bi] << 20) / 16, false);
                            rd(p2 == 1, pool, n16, true);
                            CK(hipStreamSynchronize(s0));
                            CK(hipMemcpy(hst.data(), stamps, 256 * 16, hipMemcpyDeviceToHost));
                            const double us = span_us(hst, 256);
                            best = std::min(best, us); sum += us;
                        }
                        printf("  S %6.2f MB  first %-7s second %-7s between %3d MB : %7.2f us best %7.2f avg  %6.2f TB/s (best)\n", mb,
                               p1 == 0 ? "none" : p1 == 1 ? "default" : "nt", p2 ? "nt" : "default", betweens[bi], best, sum / reps, mb * 1e6 / best / 1e6);
                    }
                }
            }
        }
        CK(hipFree(pool));
    }

    // ------------------------------------------------------------ B / C
    const double mb[4] = {31.46, 10.49, 56.36, 28.18};
    // order inside a layer: QKV, [attention], Wo, W1W3, W2
    std::vector<Seg> segs;      // weight kernels only, gate = index among weight kernels
    size_t off = 0;
    for (int l = 0; l < layers; l++)
        for (int k = 0; k < 4; k++) {
            const size_t n16 = ((size_t)(mb[k] * 1e6 / 16) + 255) / 256 * 256;
            segs.push_back({off, n16, (int)segs.size()});
            off += n16;
        }
    printf("B. chain: %d layers, %.2f GB of weights\n", layers, off * 16 / 1e9);
    u32x4 *W; CK(hipMalloc(&W, off * 16)); CK(hipMemset(W, 0x11, off * 16));
    Seg *dsegs; CK(hipMalloc(&dsegs, segs.size() * sizeof(Seg))); CK(hipMemcpy(dsegs, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
    unsigned *x0, *x1, *prog, *err;
    CK(hipMalloc(&x0, XN * 2)); CK(hipMalloc(&x1, XN * 2)); CK(hipMemset(x0, 1, XN * 2)); CK(hipMemset(x1, 1, XN * 2));
    CK(hipMalloc(&prog, 64)); CK(hipMalloc(&err, 64)); CK(hipMemset(err, 0, 64));
    const int nk = layers * 5;
    unsigned long long *tl; CK(hipMalloc(&tl, (size_t)nk * 32)); CK(hipMemset(tl, 0, (size_t)nk * 32));
    unsigned long long *pst; CK(hipMalloc(&pst, 64 * 8)); CK(hipMemset(pst, 0, 64 * 8));
    hipEvent_t t0, t1, ev; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreate(&ev));

    // chain graph; attn_prefetch_mb > 0: the attention kernel's surplus workgroups read that much of what follows it
    auto build = [&](bool nt, double attn_prefetch_mb, size_t touch_stride_bytes = 0) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        int ki = 0;
        for (int l = 0; l < layers; l++) {
            for (int k = 0; k < 5; k++, ki++) {
                MainArgs a{};
                a.x = (ki & 1) ? x1 : x0; a.y = (ki & 1) ? x0 : x1; a.prog = prog; a.stamps = tl + (size_t)ki * 4;
                if (k == 1) {
                    const Seg &nx = segs[l * 4 + 1];          // Wo, then W1W3 behind it in memory
                    a.prog = nullptr;
                    a.P = W + nx.off16; a.pn16 = (size_t)(attn_prefetch_mb * 1e6 / 16);
                    const int wgs = attn_prefetch_mb > 0 ? 256 : 32;
                    hipLaunchKernelGGL(k_attn, dim3(wgs), dim3(256), 0, s0, a, 32);
                } else {
                    const int si = l * 4 + (k == 0 ? 0 : k - 1);
                    const Seg &sg = segs[si];
                    a.W = W + sg.off16; a.n16 = sg.n16;
                    if (touch_stride_bytes && si + 1 < (int)segs.size()) { a.T = W + segs[si + 1].off16; a.tn16 = segs[si + 1].n16; a.tstride16 = touch_stride_bytes / 16; }
                    if (nt) hipLaunchKernelGGL((k_main<true, 14>), dim3(256), dim3(1024), 0, s0, a);
                    else hipLaunchKernelGGL((k_main<false, 14>), dim3(256), dim3(1024), 0, s0, a);
                }
            }
        }
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        return ge;
    };
    auto timeline = [&](const char *what) {
        std::vector<unsigned long long> h((size_t)nk * 4); CK(hipMemcpy(h.data(), tl, (size_t)nk * 32, hipMemcpyDeviceToHost));
        const int l0 = layers > 4 ? 3 : 0;
        const unsigned long long z = h[(size_t)l0 * 5 * 4];
        printf("   %s, layer %d (us from its first kernel's start; start / x arrived / end):", what, l0);
        for (int k = 0; k < 6 && l0 * 5 + k < nk; k++) {
            const unsigned long long *p = &h[(size_t)(l0 * 5 + k) * 4];
            printf("  k%d %.2f/%.2f/%.2f", k, (double)(long long)(p[0] - z) / 100.0, (double)(long long)(p[1] - z) / 100.0, (double)(long long)(p[2] - z) / 100.0);
        }
        printf("\n");
    };
    const int reps = 10;
    auto run = [&](const char *name, hipGraphExec_t ge, int pf_threads, int ahead) {
        CK(hipMemset(prog, 0, 64)); CK(hipMemset(err, 0, 64));
        unsigned base = 0;
        auto once = [&]() {
            if (pf_threads) {
                // the prefetcher of this replay may start as soon as the previous replay's chain has started
                if (pf_threads == 64) hipLaunchKernelGGL((k_prefetch<64>), dim3(256), dim3(64), 0, s1, W, dsegs, (int)segs.size(), prog, base, ahead, err, sink, pst);
                else if (pf_threads == 128) hipLaunchKernelGGL((k_prefetch<128>), dim3(256), dim3(128), 0, s1, W, dsegs, (int)segs.size(), prog, base, ahead, err, sink, pst);
                else hipLaunchKernelGGL((k_prefetch<256>), dim3(256), dim3(256), 0, s1, W, dsegs, (int)segs.size(), prog, base, ahead, err, sink, pst);
            }
            CK(hipGraphLaunch(ge, s0));
            base += (unsigned)segs.size();
        };
        for (int w = 0; w < 2; w++) once();
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        CK(hipEventRecord(t0, s0));
        for (int r = 0; r < reps; r++) once();
        CK(hipEventRecord(t1, s0));
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("%-58s: %8.1f us/chain  %6.2f us/layer  err=%u\n", name, ms * 1e3 / reps, ms * 1e3 / reps / layers, e);
        timeline(name);
    };
    hipGraphExec_t g_nt = build(true, 0), g_def = build(false, 0);
    run("serial, nt weight loads", g_nt, 0, 0);
    run("serial, default-policy weight loads", g_def, 0, 0);
    char nm[128];
    if (!(argc > 2 && atoi(argv[2]) == 4))
    for (int th : {64, 128, 256})
        for (int ahead : {2, 4, 8}) {
            snprintf(nm, sizeof nm, "prefetcher %3d thr/CU, %d kernels ahead, chain nt", th, ahead);
            run(nm, g_nt, th, ahead);
        }
    if (!(argc > 2 && atoi(argv[2]) == 4)) {
    run("prefetcher 128 thr/CU, 4 kernels ahead, chain default", g_def, 128, 4);
    run("prefetcher 128 thr/CU, unthrottled (1000 ahead), chain nt", g_nt, 128, 1000);
    }
    if (argc > 2 && atoi(argv[2]) == 4) {
        printf("D. address-translation warm-up: each kernel's workgroups 0-7 touch one line per stride of the next kernel's weights\n");
        for (size_t st : {(size_t)4096, (size_t)65536, (size_t)(2 << 20)}) {
            hipGraphExec_t g = build(true, 0, st);
            snprintf(nm, sizeof nm, "touch stride %zu B, chain nt", st);
            run(nm, g, 0, 0);
        }
        run("serial again, nt weight loads", g_nt, 0, 0);
        return 0;
    }
    printf("C. prefetch by the attention kernel's surplus workgroups\n");
    for (double pmb : {10.49, 30.0, 50.0, 66.85}) {
        hipGraphExec_t g = build(true, pmb);
        snprintf(nm, sizeof nm, "attention prefetches %.1f MB, chain nt", pmb);
        run(nm, g, 0, 0);
        hipGraphExec_t g2 = build(false, pmb);
        snprintf(nm, sizeof nm, "attention prefetches %.1f MB, chain default", pmb);
        run(nm, g2, 0, 0);
    }
    return 0;
}
