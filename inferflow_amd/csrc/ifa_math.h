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

// Greedy top-1 scan of one thread over its share of n logits: (value descending, index ascending) is a total order, so the
// scan order is free -- 16-byte requests, four in flight, instead of a 2-byte load per iteration behind a branch (32 serial
// L2 round trips: 13 us for 32000 logits).  Excluded ids (e0..e2, -1 = none) are never candidates.
typedef uint32_t am_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 am_h8 __attribute__((ext_vector_type(8)));
// id_offset: the global id of v[0] (a vocabulary shard); ids, exclusions and the result are global ids
__device__ __forceinline__ void argmax_scan(const half_t *__restrict__ v, size_t n, int e0, int e1, int e2, int tid, int nthreads,
                                            float &best, int &besti, int id_offset = 0)
{
    const size_t chunks = ((reinterpret_cast<uintptr_t>(v) & 15) == 0) ? (n >> 3) : 0;
    for (size_t c0 = (size_t)tid; c0 < chunks; c0 += (size_t)4 * nthreads) {
        am_u32x4 r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t cu = c0 + (size_t)u * nthreads;
            r[u] = reinterpret_cast<const am_u32x4 *>(v)[cu < chunks ? cu : chunks - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t cu = c0 + (size_t)u * nthreads;
            const am_h8 h = __builtin_bit_cast(am_h8, r[u]);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int i = id_offset + (int)(cu * 8) + e;
                const float f = (float)h[e];
                const bool ok = cu < chunks && i != e0 && i != e1 && i != e2;
                if (ok && (f > best || (f == best && i < besti))) { best = f; besti = i; }
            }
        }
    }
    for (size_t i = chunks * 8 + (size_t)tid; i < n; i += (size_t)nthreads) {
        const int gid = id_offset + (int)i;
        if (gid == e0 || gid == e1 || gid == e2) continue;
        const float f = h2f(v[i]);
        if (f > best || (f == best && gid < besti)) { best = f; besti = gid; }
    }
}

} // namespace ifa
