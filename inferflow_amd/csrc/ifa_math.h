// ifa_math.h -- scalar device functions shared by the op-level kernels and the
// fused decode kernels, so both paths round identically.
#pragma once
#include "ifa_device.h"

namespace ifa {

// Tensor_RmsNorm_Kernel (src/kernels/unary_tensor_opr.h:216-289): thread `tid`
// of `nthreads` sums its contiguous chunk with `sum += (double)x*(double)x` on a
// float accumulator.  For F16 inputs that is bit-identical to an fp32 fma chain:
// x*x is exact in fp32 (11-bit significands), and the double-precision add of
// two floats followed by rounding to float equals the correctly rounded fp32 add
// (exact in double when the exponents are within 29 bits; otherwise the addend is
// far below half an ulp and both round to the larger operand).  The oracle keeps
// the double formulation; tests/ check the two agree bit for bit.
__device__ __forceinline__ float rms_partial(const half_t *src, int cols, int tid, int nthreads)
{
    const int x_len = (cols + nthreads - 1) / nthreads;
    const int xs0 = tid * x_len, xe = min((tid + 1) * x_len, cols);
    float sum = 0.0f;
    if ((x_len & 7) == 0 && xe - xs0 == x_len && ((reinterpret_cast<uintptr_t>(src + xs0) & 15) == 0)) {          // vector loads first, then the ordered chain
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        for (int xi = xs0; xi < xe; xi += 8) {
            const h8 v = *reinterpret_cast<const h8 *>(src + xi);
#pragma unroll
            for (int i = 0; i < 8; i++) sum = __builtin_fmaf((float)v[i], (float)v[i], sum);
        }
        return sum;
    }
    for (int xi = xs0; xi < xe; xi++) {
        const float v = h2f(src[xi]);
        sum = __builtin_fmaf(v, v, sum);
    }
    return sum;
}

__device__ __forceinline__ float rms_scale_from_partials(const float *part, int n, int cols, float eps)
{
    // strictly ordered sum (thread 0 of the reference kernel); reads are issued in
    // batches of 64 values so the LDS latency is paid twice, not once per element
    float total = 0.0f;
    int i0 = 0;
    if ((reinterpret_cast<uintptr_t>(part) & 15) == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (; i0 + 64 <= n; i0 += 64) {
            f4 buf[16];
#pragma unroll
            for (int i = 0; i < 16; i++) buf[i] = reinterpret_cast<const f4 *>(part + i0)[i];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                total = total + buf[i][0]; total = total + buf[i][1];
                total = total + buf[i][2]; total = total + buf[i][3];
            }
        }
    }
    for (int i = i0; i < n; i++) total = total + part[i];
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
