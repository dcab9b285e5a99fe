// ifa_math.h -- scalar device functions shared by the op-level kernels and the
// fused decode kernels, so both paths round identically.
#pragma once
#include "ifa_device.h"

namespace ifa {

// RMS statistics.  Tensor_RmsNorm_Kernel (src/kernels/unary_tensor_opr.h:216-289) sums x*x over 128 strided per-thread
// chunks and lets thread 0 add the 128 partials one after the other; the value is mean(x^2) in fp32 either way, only
// the ORDER of the fp32 additions is the reference's launch geometry.  Every kernel here (op level, fused decode
// prologues, lm_head) uses ONE order that needs no serial chain and no staging barrier -- so all paths stay bit-identical
// to each other -- and the parity tests compare with the oracle's restated reference order within one half ulp of the
// normalised value (tests/test_gpu_ops.py states the bound):
//   p_c   = fma chain over the 8 elements of chunk c, ascending, from 0          (rms_chunk_sq)
//   P_g   = wave butterfly sum of p_{64g} .. p_{64g+63}                           (wave_sum: lane = c % 64)
//   total = ((0 + P_0) + P_1) + ...  ascending g                                  (rms_total)
// Missing elements / chunks count as zeros (adding +0 is exact).
typedef _Float16 rms_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float rms_chunk_sq(const rms_h8 v)
{
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) s = __builtin_fmaf((float)v[i], (float)v[i], s);
    return s;
}

__device__ __forceinline__ float rms_total(const float *group_sums, int ngroups)
{
    float total = 0.0f;
    for (int g = 0; g < ngroups; g++) total = total + group_sums[g];
    return total;
}

__device__ __forceinline__ float rms_scale_of(float total, int cols, float eps)
{
    float mean = total / (float)cols;
    return 1.0f / sqrtf(mean + eps);
}

__device__ __forceinline__ float rms_apply(float x, float scale, const half_t *w, const half_t *b, float multi_base)
{
    float v = x * scale;
    if (w) {
        float m = multi_base + h2f(*w);
        v = v * m;
        if (b) v = v + h2f(*b);
    }
    return v;
}

// RoPE on one (col) pair of one head row; order 2: (col, col+rope_cols/2),
// order 1: (2col, 2col+1).  unary_tensor_opr.h:661-740.
__device__ __forceinline__ void rope_angle(int col_for_pow, int pos, float theta, int rope_dims, float &c, float &s)
{
    const float theta_scale = powf(theta, -2.0f / (float)rope_dims);
    float ang = (float)pos;
    if (col_for_pow > 0) ang *= powf(theta_scale, (float)col_for_pow);
    c = cosf(ang); s = sinf(ang);
}

__device__ __forceinline__ void rope_rotate(half_t *row, int col, int pos, float theta, int order, int rope_dims,
                                            int rope_cols)
{
    float c, s;
    if (order == 2) {
        if (2 * col >= rope_cols) return;
        rope_angle(col, pos, theta, rope_dims, c, s);
        const float x0 = h2f(row[col]), x1 = h2f(row[col + rope_cols / 2]);
        float a = x0 * c, bq = x1 * s, d = x0 * s, e = x1 * c;
        row[col] = f2h(a - bq);
        row[col + rope_cols / 2] = f2h(d + e);
    } else {
        rope_angle(col, pos, theta, rope_dims, c, s);
        const float x0 = h2f(row[2 * col]), x1 = h2f(row[2 * col + 1]);
        float a = x0 * c, bq = x1 * s, d = x0 * s, e = x1 * c;
        row[2 * col] = f2h(a - bq);
        row[2 * col + 1] = f2h(d + e);
    }
}

// PosEmbedding_Alibi_Std_Kernel (unary_tensor_opr.h:742-762)
__device__ __forceinline__ float alibi_slope(int head, int total_heads)
{
    const int hl2 = 1 << (int)floorf(log2f((float)total_heads));
    const float m0 = powf(2.0f, -8.0f / (float)hl2);
    const float m1 = powf(2.0f, -4.0f / (float)hl2);
    return head < hl2 ? powf(m0, (float)(head + 1)) : powf(m1, (float)(2 * (head - hl2) + 1));
}

// SiluActivation_Kernel :552-576, GeluActivation_Kernel :578-594, ReluActivation_Kernel :537-550
__device__ __forceinline__ float act_fn(float v, int kind)
{
    if (kind == 0) return v / (1.0f + expf(-v));
    if (kind == 1) {
        const float GELU_COEF_A = 0.044715f;
        const float SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
        float v2 = v * v;
        float inner = 1.0f + GELU_COEF_A * v2;
        float a = SQRT_2_OVER_PI * v;
        a = a * inner;
        float t = 1.0f + tanhf(a);
        float fx = 0.5f * v;
        return fx * t;
    }
    return v > 0 ? v : 0.0f;
}

} // namespace ifa
