// ifa_device.h -- device-side helpers shared by all gfx950 kernels.
// wave64 only; no CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ifa {

typedef _Float16 half_t;

// dtype ids == reference ElementType enum (src/tensor/tensor_common.h:15-42)
enum DType : int {
    F32 = 0, F16 = 1,
    Q8_B32T1 = 7, Q8_B32T2 = 8, Q6_B64T1 = 9, Q5_B64T1 = 10, Q5_B32T1 = 11,
    Q4_B16 = 12, Q4_B32T1A = 13, Q4_B32T1B = 14, Q4_B64T1 = 17, Q3H_B64T1 = 18,
    Q3_B32T1A = 19, Q3_B32T1B = 20, Q2_B32T1A = 21, Q2_B32T1B = 22
};

// internal id (never crosses the C ABI): Q3H_B64T1 values streamed at their native 32 bytes per block by the fused decode GEMV
// (ifa_decode_formats.h WRowQ3HN; engine option q3h_native) instead of the 36-byte nibble-pair form
constexpr int Q3H_NATIVE = 99;

__host__ __device__ constexpr int block_capacity(int dt)
{
    return (dt == F32 || dt == F16) ? 1
        : (dt == Q4_B16) ? 16
        : (dt == Q6_B64T1 || dt == Q5_B64T1 || dt == Q4_B64T1 || dt == Q3H_B64T1 || dt == Q3H_NATIVE) ? 64
        : (dt == Q8_B32T1 || dt == Q8_B32T2 || dt == Q5_B32T1 || dt == Q4_B32T1A || dt == Q4_B32T1B
           || dt == Q3_B32T1A || dt == Q3_B32T1B || dt == Q2_B32T1A || dt == Q2_B32T1B) ? 32 : 0;
}

__host__ __device__ constexpr int block_bytes(int dt)
{
    return dt == F32 ? 4 : dt == F16 ? 2
        : dt == Q8_B32T1 ? 36 : dt == Q8_B32T2 ? 34 : dt == Q6_B64T1 ? 52 : dt == Q5_B64T1 ? 44
        : dt == Q5_B32T1 ? 24 : dt == Q4_B16 ? 10 : (dt == Q4_B32T1A || dt == Q4_B32T1B) ? 20
        : dt == Q4_B64T1 ? 36 : (dt == Q3H_B64T1 || dt == Q3H_NATIVE) ? 32 : (dt == Q3_B32T1A || dt == Q3_B32T1B) ? 16
        : (dt == Q2_B32T1A || dt == Q2_B32T1B) ? 12 : 0;
}

__host__ __device__ constexpr bool ax8_eligible(int dt)
{   // GetUseFullQuantGemv, src/transformer/inference_worker.cc:2707-2730
    return dt == Q8_B32T2 || dt == Q6_B64T1 || dt == Q5_B64T1 || dt == Q4_B32T1A || dt == Q4_B32T1B
        || dt == Q4_B64T1 || dt == Q3H_B64T1;
}

// ---- fp16 helpers (RN conversions, like CUDA __float2half_rn) -------------
__device__ __forceinline__ float h2f(half_t h) { return (float)h; }
// The empty asm keeps the fp32 value opaque so the compiler cannot fold the
// producing fmul/fadd and the conversion into v_fma_mixlo_f16, which rounds
// once from the exact result; the reference rounds to fp32 first, then to fp16.
__device__ __forceinline__ half_t f2h(float f) { asm volatile("" : "+v"(f)); return (half_t)f; }
__device__ __forceinline__ float hbits2f(uint16_t b) { return (float)__builtin_bit_cast(half_t, b); }
// byte B of a word as a float: ONE v_cvt_f32_ubyteN (the compiler extracts the byte first and converts with ubyte0:
// two instructions per value, measured in the dequantising GEMM kernels where they are a fifth of the VALU stream)
template <int B>
__device__ __forceinline__ float ubyte_f32(uint32_t w)
{
    float f;
    if constexpr (B == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(w));
    else if constexpr (B == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(w));
    else if constexpr (B == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(w));
    return f;
}
__device__ __forceinline__ uint16_t f2hbits(float f) { return __builtin_bit_cast(uint16_t, f2h(f)); }

// ---- wave64 reductions -----------------------------------------------------
// DPP within 16-lane rows, ds_bpermute across rows.  All lanes get the result.
template <typename T>
__device__ __forceinline__ T dpp_xor1(T v) { return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false); }
template <typename T>
__device__ __forceinline__ T dpp_xor2(T v) { return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false); }
template <typename T>
__device__ __forceinline__ T dpp_half_mirror(T v) { return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false); }
template <typename T>
__device__ __forceinline__ T dpp_mirror(T v) { return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false); }

__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    v += dpp_half_mirror(v);
    v += dpp_mirror(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_xor1(v));
    v = fmaxf(v, dpp_xor2(v));
    v = fmaxf(v, dpp_half_mirror(v));
    v = fmaxf(v, dpp_mirror(v));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// max over each aligned group of 32 lanes (a Q8_B32T2 block == 32 lanes)
__device__ __forceinline__ float half_wave_max(float v)
{
    v = fmaxf(v, dpp_xor1(v));
    v = fmaxf(v, dpp_xor2(v));
    v = fmaxf(v, dpp_half_mirror(v));
    v = fmaxf(v, dpp_mirror(v));
    v = fmaxf(v, __shfl_xor(v, 16));
    return v;
}

// sum over each aligned group of 32 lanes (exact: integers)
__device__ __forceinline__ int half_wave_sum_i32(int v)
{
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    v += dpp_half_mirror(v);
    v += dpp_mirror(v);
    v += __shfl_xor(v, 16);
    return v;
}

// 8 half x 8 half products accumulated in fp32 (products of halfs are exact in fp32)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8v_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float dot8_f16(u32x4_t w, u32x4_t x, float acc)
{
    const half8v_t wh = __builtin_bit_cast(half8v_t, w);
    const half8v_t xh = __builtin_bit_cast(half8v_t, x);
#pragma unroll
    for (int i = 0; i < 8; i++) acc = __builtin_fmaf((float)wh[i], (float)xh[i], acc);
    return acc;
}

__device__ __forceinline__ int sdot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

__device__ __forceinline__ uint32_t ld_u32_unaligned2(const uint8_t *p)
{   // 2-byte aligned 32-bit read
    const uint16_t *q = reinterpret_cast<const uint16_t *>(p);
    return (uint32_t)q[0] | ((uint32_t)q[1] << 16);
}

} // namespace ifa
