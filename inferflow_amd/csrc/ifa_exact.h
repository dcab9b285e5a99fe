// ifa_exact.h -- single-token ops in the reference kernels' summation order (csrc/ifa_exact.hip); used by the worker's
// `exact_order` option (csrc/ifa_engine_exact.hip).
#pragma once
#include "ifa_host.h"
#include "ifa_device.h"

namespace ifa {

int exact_init(hipStream_t s);
int exact_rmsnorm(const half_t *x, int rows, int cols, const half_t *w, const half_t *b, float multi_base, float eps, half_t *y, hipStream_t s);
int exact_gemv_ax8(int w_dtype, const void *W_aos, size_t rows, size_t cols, const void *xq8, const half_t *bias, half_t *y, hipStream_t s);
int exact_gemv_f16x(int w_dtype, const void *W_aos, size_t rows, size_t cols, const half_t *x, const half_t *bias, half_t *y, hipStream_t s);
int exact_rope(half_t *x, int head_dim, int heads, const float *tab_row, int order, hipStream_t s);
int exact_attention(const half_t *q, const void *kc, const void *vc, int kv_dtype, size_t row_bytes, int n_ctx, int heads, int kv_heads, int head_dim,
                    float alpha, float sm_scale, half_t *out, hipStream_t s);
int exact_softmax_row(half_t *s_row, int cx, float scale, hipStream_t s);
int exact_moe_combine(half_t *f, const half_t *expert_out, const half_t *weight_dev, size_t n, hipStream_t s);
int exact_act_mul(int kind, const half_t *a, const half_t *gate, size_t n, half_t *y, hipStream_t s);

} // namespace ifa
