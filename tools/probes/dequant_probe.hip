// dequant_probe.hip -- q4x8_dequant (ifa_dequant_q4.h: FP8 read of the nibble, two codes per conversion) against the plain
// v_cvt_f32_ubyte form for random (scale, base) halves and all code words of a few patterns: the halves must be bit-identical.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I inferflow_amd/csrc -o dequant_probe dequant_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "ifa_dequant_q4.h"
using namespace ifa;

__global__ void k_new(const uint32_t *cw, const uint16_t *sb, q4_h2 *out_new, float *unit, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float up = q4_fp8_up();
    if (i == 0) unit[0] = 1.0f / up;
    const float scale = (float)__builtin_bit_cast(_Float16, sb[2 * i]), base = (float)__builtin_bit_cast(_Float16, sb[2 * i + 1]);
    q4_h2 w[4];
    q4x8_dequant(cw[i], scale * up, base, w);
    for (int p = 0; p < 4; p++) out_new[i * 4 + p] = w[p];
}
__global__ void k_ref(const uint32_t *cw, const uint16_t *sb, uint16_t *out_ref, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float scale = (float)__builtin_bit_cast(_Float16, sb[2 * i]), base = (float)__builtin_bit_cast(_Float16, sb[2 * i + 1]);
    for (int e = 0; e < 8; e++) {
        const uint32_t q = (cw[i] >> (8 * (e / 2) + 4 * (e & 1))) & 15u;
        float f = __builtin_fmaf((float)q, scale, base);
        asm volatile("" : "+v"(f));
        out_ref[i * 8 + e] = __builtin_bit_cast(uint16_t, (_Float16)f);
    }
}

int main()
{
    const int n = 1 << 20;
    std::vector<uint32_t> cw(n); std::vector<uint16_t> sb(2 * n);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int i = 0; i < n; i++) {
        cw[i] = (uint32_t)rnd();
        uint16_t s = (uint16_t)rnd(), b = (uint16_t)rnd();
        if ((s & 0x7C00) == 0x7C00) s &= 0x3FFF;       // no inf / nan
        if ((b & 0x7C00) == 0x7C00) b &= 0x3FFF;
        if (i & 1) { s = (s & 0x83FF) | 0x2000; b = (b & 0x83FF) | 0x2C00; }       // typical magnitudes too
        sb[2 * i] = s; sb[2 * i + 1] = b;
    }
    uint32_t *dcw; uint16_t *dsb, *dn, *dr; float *du;
    hipMalloc(&dcw, n * 4); hipMalloc(&dsb, n * 4); hipMalloc(&dn, n * 16); hipMalloc(&dr, n * 16); hipMalloc(&du, 4);
    hipMemcpy(dcw, cw.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), n * 4, hipMemcpyHostToDevice);
    k_new<<<n / 256, 256>>>(dcw, dsb, (q4_h2 *)dn, du, n);
    k_ref<<<n / 256, 256>>>(dcw, dsb, dr, n);
    std::vector<uint16_t> a(n * 8), b(n * 8); float unit = 0;
    hipMemcpy(a.data(), dn, n * 16, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dr, n * 16, hipMemcpyDeviceToHost); hipMemcpy(&unit, du, 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (long i = 0; i < (long)n * 8; i++) {
        const bool nan_a = (a[i] & 0x7FFF) > 0x7C00, nan_b = (b[i] & 0x7FFF) > 0x7C00;
        if (a[i] != b[i] && !(nan_a && nan_b)) { if (bad < 5) printf("mismatch at %ld: new %04x ref %04x (cw %08x scale %04x base %04x)\n", i, a[i], b[i], cw[i / 8], sb[2 * (i / 8)], sb[2 * (i / 8) + 1]); bad++; }
    }
    printf("fp8 code 1 = %g (2^-9 = %g); %ld mismatches of %ld dequantised halves\n", unit, 1.0 / 512, bad, (long)n * 8);
    return bad != 0;
}
