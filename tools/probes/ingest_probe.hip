// ingest_probe.hip -- how fast ONE compute unit pulls an L2-resident (or HBM-resident) buffer, by request form:
//   A: ordinary 16-byte loads into registers (8 in flight per lane), B: direct-to-LDS loads (global_load_lds_dwordx4) into a ring,
// for 1 / 2 workgroups of 256 threads per CU and a buffer of 1 MB (every workgroup reads the SAME bytes: L2 after the first touch)
// or 4 GB / workgroups (every workgroup its own bytes: HBM).  Prints GB/s per CU.
// hipcc --offload-arch=gfx950 -O3 -o ingest_probe ingest_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_t;
typedef const __attribute__((address_space(1))) void glb_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// each workgroup walks `bytes` bytes from its base `reps` times in steps of 256 threads x 16 B x U
template <int U>
__global__ void __launch_bounds__(256) k_loads(const uint8_t *base, size_t wg_stride, size_t bytes, int reps, uint32_t *sink)
{
    const uint8_t *p = base + (size_t)blockIdx.x * wg_stride + (size_t)threadIdx.x * 16;
    u4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; r++)
        for (size_t o = 0; o + (size_t)U * 4096 <= bytes; o += (size_t)U * 4096) {
            u4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const u4 *>(p + o + (size_t)u * 4096);
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= v[u];
        }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

// NS stages of SB bytes per workgroup; every wave requests its quarter of a stage (SB / 4 / 1024 instructions), NS - 1 stages ahead
template <int NS, int SB>
__global__ void __launch_bounds__(256) k_dma(const uint8_t *base, size_t wg_stride, size_t bytes, int reps, uint32_t *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PW = SB / 4 / 1024;            // 1 KB requests per wave and stage
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint8_t *p = base + (size_t)blockIdx.x * wg_stride + (size_t)wave * (SB / 4) + (size_t)lane * 16;
    const size_t nst = bytes / SB * (size_t)reps, per = bytes / SB;
    auto issue = [&](size_t st, int slot) {
        const size_t o = (st % per) * SB;
#pragma unroll
        for (int j = 0; j < PW; j++)
            __builtin_amdgcn_global_load_lds((glb_t *)(p + o + (size_t)j * 1024), (lds_t *)(smem + (size_t)slot * SB + (size_t)wave * (SB / 4) + (size_t)j * 1024), 16, 0, 0);
    };
    for (int s = 0; s < NS - 1; s++) issue(s, s);
    int slot = 0;
    uint32_t acc = 0;
    for (size_t st = 0; st < nst; st++) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * PW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(st + NS - 1 < nst ? st + NS - 1 : nst - 1, slot == 0 ? NS - 1 : slot - 1);
        uint32_t v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(slot * SB + threadIdx.x * 4)));
        acc ^= v;
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = 1;
}

template <typename F>
static float time_ms(F launch, hipStream_t s)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < 5; i++) launch();
    (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t big = (size_t)4 << 30;
    uint8_t *buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
    uint32_t *sink; CK(hipMalloc(&sink, 64));
    for (int per_cu = 1; per_cu <= 2; per_cu++) {
        const int wgs = 256 * per_cu;
        for (int hbm = 0; hbm <= 1; hbm++) {
            const size_t bytes = hbm ? big / wgs / 4096 * 4096 / 8 : (size_t)1 << 20;      // per workgroup and repetition
            const size_t stride = hbm ? big / wgs / 4096 * 4096 : 0;
            const int reps = hbm ? 1 : 16;
            const double total = (double)bytes * reps;
            auto rep = [&](const char *name, float ms) {
                printf("%-34s %d WG/CU  %s : %7.1f GB/s per CU   (%.1f TB/s chip)\n", name, per_cu, hbm ? "HBM (own bytes)" : "L2  (same 1 MB)", total * per_cu / (ms * 1e-3) / 1e9, total * wgs / (ms * 1e-3) / 1e12);
            };
            rep("loads to registers, 4 in flight", time_ms([&] { k_loads<4><<<wgs, 256, 0, s>>>(buf, stride, bytes, reps, sink); }, s));
            rep("loads to registers, 8 in flight", time_ms([&] { k_loads<8><<<wgs, 256, 0, s>>>(buf, stride, bytes, reps, sink); }, s));
            rep("loads to registers, 16 in flight", time_ms([&] { k_loads<16><<<wgs, 256, 0, s>>>(buf, stride, bytes, reps, sink); }, s));
            rep("direct-to-LDS, 3 x 16 KB ring", time_ms([&] { k_dma<3, 16384><<<wgs, 256, 3 * 16384, s>>>(buf, stride, bytes, reps, sink); }, s));
            (void)hipFuncSetAttribute((const void *)k_dma<4, 16384>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
            rep("direct-to-LDS, 4 x 16 KB ring", time_ms([&] { k_dma<4, 16384><<<wgs, 256, 4 * 16384, s>>>(buf, stride, bytes, reps, sink); }, s));
            (void)hipFuncSetAttribute((const void *)k_dma<3, 32768>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
            if (per_cu == 1) rep("direct-to-LDS, 3 x 32 KB ring", time_ms([&] { k_dma<3, 32768><<<wgs, 256, 3 * 32768, s>>>(buf, stride, bytes, reps, sink); }, s));
        }
    }
    return 0;
}
