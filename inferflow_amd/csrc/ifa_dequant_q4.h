// ifa_dequant_q4.h -- the dequantised halves of eight 4-bit codes, value = half(fma(q, scale, base)) in fp32 (the reference's
// dequantisation kernels: one fp32 fma per weight, one rounding to F16), with the nibble -> float conversion TWO codes per
// instruction: a byte 0..15 read as an FP8 E4M3 number is b * u exactly (codes 0..7 are its subnormals m * 2^(1 - bias - 3),
// codes 8..15 the first binade (8 + m) * 2^(1 - bias - 3): one linear ramp; u = 2^-9 for the OCP format of gfx950, 2^-10 for
// FNUZ -- read from the hardware once per kernel, q4_fp8_up()), so v_cvt_pk_f32_fp8 replaces two v_cvt_f32_ubyteN and
// fma(b * u, scale / u, base) is the same real product and sum as fma(b, scale, base): one rounding, identical bits (scale / u
// is exact: u is a power of two and scale a half).  15 VALU instructions per 8 weights instead of 19.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace ifa {

typedef float q4_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 q4_h2 __attribute__((ext_vector_type(2)));

// 1 / (value of FP8 code 1): multiply the block's scale by this once
__device__ __forceinline__ float q4_fp8_up()
{
    const q4_f2 one = __builtin_amdgcn_cvt_pk_f32_fp8(0x0101, false);
    return 1.0f / one[0];
}

// code word cw: byte b holds element 2b (low nibble) and 2b + 1 (high nibble) of the word's eight weights.
// w[i] = (element 2i, element 2i + 1) as packed halves; scale_up = scale * q4_fp8_up()
__device__ __forceinline__ void q4x8_dequant(uint32_t cw, float scale_up, float base, q4_h2 (&w)[4])
{
    const uint32_t lo = cw & 0x0F0F0F0Fu, hi = (cw >> 4) & 0x0F0F0F0Fu;
    const q4_f2 s2 = {scale_up, scale_up}, b2 = {base, base};
    const q4_f2 e02 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), s2, b2);
    const q4_f2 e46 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true), s2, b2);
    const q4_f2 o13 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), s2, b2);
    const q4_f2 o57 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true), s2, b2);
    w[0] = __builtin_convertvector(q4_f2{e02[0], o13[0]}, q4_h2);
    w[1] = __builtin_convertvector(q4_f2{e02[1], o13[1]}, q4_h2);
    w[2] = __builtin_convertvector(q4_f2{e46[0], o57[0]}, q4_h2);
    w[3] = __builtin_convertvector(q4_f2{e46[1], o57[1]}, q4_h2);
}

} // namespace ifa
