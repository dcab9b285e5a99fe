// valu_rate_probe.hip -- issue cost (cycles per wave instruction) of the VALU instructions of the Q4 dequantisation on gfx950:
// v_cvt_pk_f32_fp8, v_pk_fma_f32, v_cvt_pk_f16_f32, v_cvt_f32_ubyteN, v_dot2c_f32_f16, and the whole q4x8_dequant (15
// instructions per 8 weights) -- one wave per SIMD and two, against a dependent v_add_f32 chain (4 cycles per instruction) that
// also gives the shader clock.   hipcc --offload-arch=gfx950 -O3 -I inferflow_amd/csrc -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "ifa_dequant_q4.h"
using namespace ifa;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(int iters, float *out, long long *ticks, uint32_t seed)
{
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    f2 p0 = {a, b}, p1 = {b, a}, p2 = {a + 1, b}, p3 = {b, a + 2};
    uint32_t w0 = seed + threadIdx.x, w1 = seed * 3 + threadIdx.x, w2 = seed * 5, w3 = seed * 7;
    q4_h2 acc_h[4] = {};
    float acc = 0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == 0) {          // dependent adds
#pragma unroll
            for (int u = 0; u < 16; u++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if constexpr (MODE == 1) {   // 16 independent v_cvt_pk_f32_fp8
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(p0) : "v"(w0));
                asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(p1) : "v"(w1));
                asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(p2) : "v"(w2));
                asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(p3) : "v"(w3));
            }
        } else if constexpr (MODE == 2) {   // 16 independent v_pk_fma_f32
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(p2), "v"(p3));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p2) : "v"(p3), "v"(p0));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p3) : "v"(p0), "v"(p1));
            }
        } else if constexpr (MODE == 3) {   // 16 v_cvt_pk_f16_f32
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w0) : "v"(a), "v"(b));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w1) : "v"(b), "v"(a));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w2) : "v"(a), "v"(a));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w3) : "v"(b), "v"(b));
            }
        } else if constexpr (MODE == 4) {   // 16 v_cvt_f32_ubyte
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(p0[0]) : "v"(w0));
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(p1[0]) : "v"(w1));
                asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(p2[0]) : "v"(w2));
                asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(p3[0]) : "v"(w3));
            }
        } else if constexpr (MODE == 5) {   // 16 v_dot2c_f32_f16 on four accumulators
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(p0[0]) : "v"(w0), "v"(w1));
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(p1[0]) : "v"(w1), "v"(w2));
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(p2[0]) : "v"(w2), "v"(w3));
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(p3[0]) : "v"(w3), "v"(w0));
            }
        } else {                             // MODE 6: two whole q4x8_dequant (30 instructions, 16 weights)
            q4_h2 w[4];
            q4x8_dequant(w0, a, b, w);
            asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
            w0 ^= __builtin_bit_cast(uint32_t, w[0]);
            q4x8_dequant(w1, b, a, w);
            asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
            w1 ^= __builtin_bit_cast(uint32_t, w[3]);
        }
    }
    const long long t1 = wall_clock64();
    acc = a + p0[0] + p1[1] + p2[0] + p3[1] + (float)(w0 ^ w1 ^ w2 ^ w3);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *what, int n_instr, float *out, long long *ticks, double ghz)
{
    for (int waves_per_simd = 1; waves_per_simd <= 2; waves_per_simd++) {
        const int iters = 20000;
        k<MODE><<<256, 256 * waves_per_simd>>>(iters, out, ticks, 12345u);
        hipDeviceSynchronize();
        long long t[256]; hipMemcpy(t, ticks, sizeof(t), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; i++) avg += t[i]; avg /= 256;
        const double ns = avg * 10.0;
        printf("%-34s %d wave(s) per SIMD: %.2f ns per loop body of %d instructions per wave = %.2f ns per instruction", what, waves_per_simd, ns / iters, n_instr, ns / iters / n_instr);
        if (ghz > 0) printf(" = %.1f cycles at %.2f GHz (SIMD: %.1f cycles per wave instruction)", ns / iters / n_instr * ghz, ghz, ns / iters / n_instr * ghz / waves_per_simd);
        printf("\n");
    }
}

int main()
{
    float *out; long long *ticks;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8);
    // clock from the dependent add chain: 4 cycles per instruction
    k<0><<<256, 256>>>(20000, out, ticks, 1u); hipDeviceSynchronize();
    k<0><<<256, 256>>>(20000, out, ticks, 1u); hipDeviceSynchronize();
    long long t[256]; hipMemcpy(t, ticks, sizeof(t), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; i++) avg += t[i]; avg /= 256;
    const double ns_per_add = avg * 10.0 / 20000 / 16;
    const double ghz = 4.0 / ns_per_add;
    printf("dependent v_add_f32: %.3f ns each -> %.2f GHz if 4 cycles\n", ns_per_add, ghz);
    run<0>("v_add_f32 (dependent)", 16, out, ticks, ghz);
    run<1>("v_cvt_pk_f32_fp8", 16, out, ticks, ghz);
    run<2>("v_pk_fma_f32", 16, out, ticks, ghz);
    run<3>("v_cvt_pk_f16_f32", 16, out, ticks, ghz);
    run<4>("v_cvt_f32_ubyteN", 16, out, ticks, ghz);
    run<5>("v_dot2c_f32_f16", 16, out, ticks, ghz);
    run<6>("2 x q4x8_dequant (30 instr + 2 xor)", 32, out, ticks, ghz);
    return 0;
}
