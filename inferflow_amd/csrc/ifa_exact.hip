// ifa_exact.hip -- the T = 1 step of the path in the SUMMATION ORDER of the reference's CUDA kernels, as the oracle restates it
// (oracle/ifa_oracle.c cites the same lines): a 32-lane walk over eight weight blocks per iteration with an xor butterfly for the
// int8 GEMV (src/kernels/gemv.h:1499-1709), 128 contiguous chunks for the RMS norm (src/kernels/unary_tensor_opr.h:216-289), serial
// fp32 dots for attention scores and P x V (Gemm_Alg2_Kernel, src/kernels/gemm.h:83-178), the 32-lane softmax
// (Tensor_SoftMax_Alg2_Kernel, unary_tensor_opr.h:480-535).  Engine option `exact_order` routes single-token steps through these
// kernels (csrc/ifa_engine_exact.hip): every F16 tensor and every int8 code of a step is then bit-identical to the oracle's, which
// is what tests/test_gpu_fullsize_oracle.py asserts through 32 layers.  They are correctness instruments -- one half wave per row,
// no layout tricks -- not the timed path; the timed kernels keep the same per-block terms in a wave64 order.
//
// Transcendentals: the CPU side of the comparison calls libm, so (a) exp is restated here in the algorithm glibc's expf uses
// (Szabolcs Nagy's exp2f-table form: N = 32 table of 2^(i/32), cubic in double; constants as published in glibc
// sysdeps/ieee754/flt-32/e_exp2f_data.c): checked in this repository's build container against libm over every float in
// [-103, 88]: 2 of 2.2e9 inputs differ (the library's FMA contraction of the double polynomial); (b) the RoPE angles come from a
// host-built table (powf / cosf / sinf of the host libm, csrc/ifa_engine_exact.hip).
#include "ifa_host.h"
#include "ifa_codec.h"
#include "ifa_exact.h"

namespace ifa {

// ---------------------------------------------------------------- expf, glibc's algorithm
__constant__ double c_exp2_tab[32];     // 2^(i/32), filled once per device by exact_init
__device__ __forceinline__ float expf_libm(float x)
{
    if (x != x) return x;
    if (x == -INFINITY) return 0.0f;
    if (x > 0x1.62e42ep6f) return INFINITY;
    if (x < -0x1.9fe368p6f) return 0.0f;
    constexpr double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    double z = InvLn2N * (double)x;
    double kd = z + SHIFT;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd = kd - SHIFT;
    const double r = z - kd;
    uint64_t t = (uint64_t)__double_as_longlong(c_exp2_tab[ki & 31]) - ((ki & 31) << 47);
    t += ki << 47;
    const double s = __longlong_as_double((long long)t);
    z = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

// ---------------------------------------------------------------- RMS norm, 128 contiguous chunks (orc_rmsnorm)
__global__ void __launch_bounds__(128) k_exact_rmsnorm(const half_t *__restrict__ x, int cols, const half_t *__restrict__ w,
                                                       const half_t *__restrict__ b, float multi_base, float eps, half_t *__restrict__ y)
{
    __shared__ float part[128];
    __shared__ float s_scale;
    const int tid = threadIdx.x;
    const half_t *src = x + (size_t)blockIdx.x * cols;
    const int x_len = (cols + 127) / 128;
    const int xs0 = tid * x_len, xe = min((tid + 1) * x_len, cols);
    float sum = 0.0f;
    for (int xi = xs0; xi < xe; xi++) {
        const double v = (double)h2f(src[xi]);
        const double sq = v * v;
        sum = (float)((double)sum + sq);
    }
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        float total = 0.0f;
        for (int i = 0; i < 128; i++) total = total + part[i];
        const float mean = total / (float)cols;
        s_scale = 1.0f / sqrtf(mean + eps);
    }
    __syncthreads();
    const float scale = s_scale;
    half_t *dst = y + (size_t)blockIdx.x * cols;
    for (int xi = tid; xi < cols; xi += 128) {
        float v = h2f(src[xi]) * scale;
        if (w) {
            const float m = multi_base + h2f(w[xi]);
            v = v * m;
            if (b) v = v + h2f(b[xi]);
        }
        dst[xi] = f2h(v);
    }
}

// ---------------------------------------------------------------- int8 GEMV, 32-lane order (orc_gemv_ax8)
// lane l of a row's 32: part = l & 3 of the blocks k = (l >> 2), (l >> 2) + 8, ...; per block-part  t = scale * (float)dot; t *= xs;
// acc += t;  u = base * (float)sum_x; u *= xs; acc += u;  then lanes[l] + lanes[l ^ mask] for mask = 16 .. 1.
template <int DT>
__global__ void __launch_bounds__(256) k_exact_gemv_ax8(const uint8_t *__restrict__ W, int rows, int nblk, const uint8_t *__restrict__ xq8,
                                                        const half_t *__restrict__ bias, half_t *__restrict__ y)
{
    constexpr int CAP = block_capacity(DT), BB = block_bytes(DT), PER = CAP == 32 ? 8 : 16;
    const int l = threadIdx.x & 31;
    const int row = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    const bool live = row < rows;
    const int part = l & 3;
    float acc = 0.0f;
    if (live) {
        const uint8_t *wrow = W + (size_t)row * nblk * BB;
        for (int k = l >> 2; k < nblk; k += 8) {
            RawBlock<BB> blk;
            blk.load(wrow + (size_t)k * BB);
            int q[CAP]; float scale, base;
            decode_block<DT>(blk, q, scale, base);
            const int e0 = part * PER;
            const int xblk = CAP == 32 ? k : 2 * k + (part >= 2 ? 1 : 0);
            const uint8_t *xb = xq8 + (size_t)xblk * 34;
            const float xs = hbits2f((uint16_t)(xb[0] | (xb[1] << 8)));
            const int xoff = CAP == 32 ? e0 : e0 - (part >= 2 ? 32 : 0);
            int gs = 0, gs2 = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int xv = (int)(int8_t)xb[2 + xoff + i];
                int qv = 0;
#pragma unroll
                for (int p = 0; p < 4; p++) qv = part == p ? q[p * PER + i] : qv;       // (q[] stays in registers: constant indices)
                gs += qv * xv; gs2 += xv;
            }
            float t = scale * (float)gs;
            t = t * xs;
            acc = acc + t;
            if constexpr (DT != Q8_B32T2) {
                float u = base * (float)gs2;
                u = u * xs;
                acc = acc + u;
            }
        }
    }
#pragma unroll
    for (int mask = 16; mask > 0; mask >>= 1) acc = acc + __shfl_xor(acc, mask, 32);
    if (live && l == 0) {
        half_t yh = f2h(acc);
        if (bias) yh = f2h(h2f(yh) + h2f(bias[row]));
        y[row] = yh;
    }
}

// ---------------------------------------------------------------- fp16-activation GEMV, serial fp32 order (orc_gemv_f16x)
template <int DT>
__global__ void __launch_bounds__(64) k_exact_gemv_f16x(const uint8_t *__restrict__ W, int rows, int cols, const half_t *__restrict__ x,
                                                        const half_t *__restrict__ bias, half_t *__restrict__ y)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    float acc = 0.0f;
    if constexpr (DT == F16) {
        const half_t *wr = reinterpret_cast<const half_t *>(W) + (size_t)row * cols;
        for (int c = 0; c < cols; c++) { const float p = h2f(wr[c]) * h2f(x[c]); acc = acc + p; }
    } else {
        constexpr int CAP = block_capacity(DT), BB = block_bytes(DT);
        const int nblk = cols / CAP;
        const uint8_t *wrow = W + (size_t)row * nblk * BB;
        for (int k = 0; k < nblk; k++) {
            RawBlock<BB> blk;
            blk.load(wrow + (size_t)k * BB);
            int q[CAP]; float scale, base;
            decode_block<DT>(blk, q, scale, base);
#pragma unroll
            for (int i = 0; i < CAP; i++) {
                const float wv = h2f(f2h(block_value<DT>(q[i], scale, base)));
                const float p = wv * h2f(x[k * CAP + i]);
                acc = acc + p;
            }
        }
    }
    half_t yh = f2h(acc);
    if (bias) yh = f2h(h2f(yh) + h2f(bias[row]));
    y[row] = yh;
}

// ---------------------------------------------------------------- RoPE from the host-built table (orc_rope)
// tab: [positions][head_dim / 2] (cos, sin); order 2: pairs (col, col + head_dim / 2), otherwise (2 col, 2 col + 1)
__global__ void __launch_bounds__(256) k_exact_rope(half_t *__restrict__ x, int head_dim, int heads, const float2 *__restrict__ tab_row, int order)
{
    const int half_hd = head_dim >> 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < heads * half_hd; i += gridDim.x * blockDim.x) {
        const int h = i / half_hd, col = i - h * half_hd;
        half_t *row = x + (size_t)h * head_dim;
        const int ia = order == 2 ? col : 2 * col, ib = order == 2 ? col + half_hd : 2 * col + 1;
        const float c = tab_row[col].x, s = tab_row[col].y;
        const float x0 = h2f(row[ia]), x1 = h2f(row[ib]);
        const float a = x0 * c, bq = x1 * s;
        const float d = x0 * s, e = x1 * c;
        row[ia] = f2h(a - bq);
        row[ib] = f2h(d + e);
    }
}

// ---------------------------------------------------------------- attention of one query row (orc_attention + orc_softmax)
// one workgroup per head; the scores / probabilities of the row live in LDS as halves, exactly the tensors the reference
// materialises between its launches
template <bool Q8>
__device__ __forceinline__ float kv_value(const uint8_t *cache, size_t row_bytes, int j, int col)
{
    if constexpr (!Q8) return h2f(reinterpret_cast<const half_t *>(cache + (size_t)j * row_bytes)[col]);
    else {
        const uint8_t *blk = cache + (size_t)j * row_bytes + (size_t)(col >> 5) * 34;
        const float scale = hbits2f((uint16_t)(blk[0] | (blk[1] << 8)));
        const int q = (int)(int8_t)blk[2 + (col & 31)];
        return h2f(f2h(block_value<Q8_B32T2>(q, scale, 0.0f)));      // the cache row dequantised to half first (kv_cache.cc:104-249)
    }
}

template <bool Q8>
__global__ void __launch_bounds__(256) k_exact_attention(const half_t *__restrict__ q, const uint8_t *__restrict__ kc, const uint8_t *__restrict__ vc,
                                                         size_t row_bytes, int n_ctx, int heads, int kv_heads, int head_dim, float alpha,
                                                         float sm_scale, half_t *__restrict__ out)
{
    extern __shared__ half_t S[];
    __shared__ float red[256];
    __shared__ float s_bcast;
    const int h = blockIdx.x, kvh = h / (heads / kv_heads), tid = threadIdx.x;
    const half_t *qv = q + (size_t)h * head_dim;
    const int col0 = kvh * head_dim;
    for (int j = tid; j < n_ctx; j += 256) {
        float c = 0.0f;
        for (int d = 0; d < head_dim; d++) { const float p = h2f(qv[d]) * kv_value<Q8>(kc, row_bytes, j, col0 + d); c = c + p; }
        S[j] = f2h(alpha * c);
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < n_ctx; j += 256) { const float v = sm_scale * h2f(S[j]); mx = mx > v ? mx : v; }
    red[tid] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] = red[tid] > red[tid + st] ? red[tid] : red[tid + st]; __syncthreads(); }
    mx = red[0];
    __syncthreads();
    if (tid < 32) {
        float lane = 0.0f;
        for (int xi = tid; xi < n_ctx; xi += 32) {
            const float v = sm_scale * h2f(S[xi]);
            const float e = expf_libm(v - mx);
            lane = lane + e;
            S[xi] = f2h(e);
        }
#pragma unroll
        for (int mask = 16; mask > 0; mask >>= 1) lane = lane + __shfl_xor(lane, mask, 32);
        if (tid == 0) s_bcast = 1.0f / lane;
    }
    __syncthreads();
    const float inv = s_bcast;
    for (int j = tid; j < n_ctx; j += 256) S[j] = f2h(h2f(S[j]) * inv);
    __syncthreads();
    for (int d = tid; d < head_dim; d += 256) {
        float c = 0.0f;
        for (int j = 0; j < n_ctx; j++) { const float p = h2f(S[j]) * kv_value<Q8>(vc, row_bytes, j, col0 + d); c = c + p; }
        out[(size_t)h * head_dim + d] = f2h(1.0f * c);
    }
}

// ---------------------------------------------------------------- activation (+ gate), orc_act then orc_mul
__global__ void __launch_bounds__(256) k_exact_act_mul(const half_t *__restrict__ a, const half_t *__restrict__ g, int n, int kind, half_t *__restrict__ y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = h2f(a[i]);
    float fx;
    if (kind == 0) fx = v / (1.0f + expf_libm(-v));
    else fx = v > 0 ? v : 0;
    half_t r = f2h(fx);
    if (g) r = f2h(h2f(r) * h2f(g[i]));
    y[i] = r;
}

// ---------------------------------------------------------------- router softmax over E experts (orc_softmax, one row) and the
// weighted accumulation of an expert's output (AddByRowIdx_Kernel as the oracle restates it: double product + sum, float, half)
__global__ void __launch_bounds__(64) k_exact_softmax_row(half_t *__restrict__ s, int cx, float scale)
{
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int xi = 0; xi < cx; xi++) { const float v = scale * h2f(s[xi]); mx = mx > v ? mx : v; }
    float lane = 0.0f;
    if (tid < 32)
        for (int xi = tid; xi < cx; xi += 32) {
            const float v = scale * h2f(s[xi]);
            const float e = expf_libm(v - mx);
            lane = lane + e;
            s[xi] = f2h(e);
        }
#pragma unroll
    for (int mask = 16; mask > 0; mask >>= 1) lane = lane + __shfl_xor(lane, mask, 32);
    const float inv = 1.0f / __shfl(lane, 0, 64);
    __syncthreads();
    if (tid < 32)
        for (int xi = tid; xi < cx; xi += 32) s[xi] = f2h(h2f(s[xi]) * inv);
}

__global__ void __launch_bounds__(256) k_exact_moe_combine(half_t *__restrict__ f, const half_t *__restrict__ eo, const half_t *__restrict__ w, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double p = (double)h2f(eo[i]) * (double)h2f(w[0]) + (double)h2f(f[i]);
    f[i] = f2h((float)p);
}

// ---------------------------------------------------------------- host entry points
int exact_init(hipStream_t s)
{
    static bool done[64] = {};
    int dev = 0;
    IFA_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && done[dev]) return IFA_OK;
    double tab[32];
    for (int i = 0; i < 32; i++) tab[i] = exp2((double)i / 32.0);
    IFA_HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_exp2_tab), tab, sizeof(tab), 0, hipMemcpyHostToDevice, s));
    IFA_HIP_CHECK(hipStreamSynchronize(s));
    if (dev >= 0 && dev < 64) done[dev] = true;
    return IFA_OK;
}

int exact_rmsnorm(const half_t *x, int rows, int cols, const half_t *w, const half_t *b, float multi_base, float eps, half_t *y, hipStream_t s)
{
    k_exact_rmsnorm<<<(unsigned)rows, 128, 0, s>>>(x, cols, w, b, multi_base, eps, y);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int exact_gemv_ax8(int w_dtype, const void *W, size_t rows, size_t cols, const void *xq8, const half_t *bias, half_t *y, hipStream_t s)
{
    IFA_REQUIRE(ax8_eligible(w_dtype), "exact GEMV: weight dtype %d is not on the int8 path", w_dtype);
    const int cap = block_capacity(w_dtype);
    IFA_REQUIRE(cols % (size_t)cap == 0, "exact GEMV: cols %zu not a multiple of %d", cols, cap);
    const unsigned grid = ifa_cdiv(rows, 8);
#define IFA_EX_CASE(T) case T: k_exact_gemv_ax8<T><<<grid, 256, 0, s>>>((const uint8_t *)W, (int)rows, (int)(cols / cap), (const uint8_t *)xq8, bias, y); break;
    switch (w_dtype) {
        IFA_EX_CASE(Q8_B32T2) IFA_EX_CASE(Q6_B64T1) IFA_EX_CASE(Q5_B64T1) IFA_EX_CASE(Q4_B32T1A) IFA_EX_CASE(Q4_B32T1B) IFA_EX_CASE(Q4_B64T1) IFA_EX_CASE(Q3H_B64T1)
    default: return ifa_fail(IFA_ERR_DTYPE, "exact GEMV: dtype %d", w_dtype);
    }
#undef IFA_EX_CASE
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int exact_gemv_f16x(int w_dtype, const void *W, size_t rows, size_t cols, const half_t *x, const half_t *bias, half_t *y, hipStream_t s)
{
    const unsigned grid = ifa_cdiv(rows, 64);
    if (w_dtype == F16) {
        k_exact_gemv_f16x<F16><<<grid, 64, 0, s>>>((const uint8_t *)W, (int)rows, (int)cols, x, bias, y);
    } else {
        const int cap = block_capacity(w_dtype);
        IFA_REQUIRE(cap > 1 && cols % (size_t)cap == 0, "exact F16-activation GEMV: dtype %d, cols %zu", w_dtype, cols);
        IFA_DISPATCH_QUANT_DTYPE(w_dtype, k_exact_gemv_f16x<DT><<<grid, 64, 0, s>>>((const uint8_t *)W, (int)rows, (int)cols, x, bias, y));
    }
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int exact_rope(half_t *x, int head_dim, int heads, const float *tab_row, int order, hipStream_t s)
{
    k_exact_rope<<<ifa_cdiv((size_t)heads * (head_dim / 2), 256), 256, 0, s>>>(x, head_dim, heads, reinterpret_cast<const float2 *>(tab_row), order);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int exact_attention(const half_t *q, const void *kc, const void *vc, int kv_dtype, size_t row_bytes, int n_ctx, int heads, int kv_heads, int head_dim,
                    float alpha, float sm_scale, half_t *out, hipStream_t s)
{
    const size_t lds = (size_t)n_ctx * 2;
    IFA_REQUIRE(lds <= 128 * 1024, "exact attention: %d keys do not fit the row buffer", n_ctx);
    IFA_REQUIRE(exact_init(s) == IFA_OK, "exact attention: table upload failed");
    if (kv_dtype == Q8_B32T2) {
        if (lds > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_exact_attention<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_exact_attention<true><<<(unsigned)heads, 256, lds, s>>>(q, (const uint8_t *)kc, (const uint8_t *)vc, row_bytes, n_ctx, heads, kv_heads, head_dim, alpha, sm_scale, out);
    } else {
        if (lds > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_exact_attention<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_exact_attention<false><<<(unsigned)heads, 256, lds, s>>>(q, (const uint8_t *)kc, (const uint8_t *)vc, row_bytes, n_ctx, heads, kv_heads, head_dim, alpha, sm_scale, out);
    }
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int exact_softmax_row(half_t *s_row, int cx, float scale, hipStream_t s)
{
    IFA_REQUIRE(exact_init(s) == IFA_OK, "exact softmax: table upload failed");
    k_exact_softmax_row<<<1, 64, 0, s>>>(s_row, cx, scale);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int exact_moe_combine(half_t *f, const half_t *expert_out, const half_t *weight_dev, size_t n, hipStream_t s)
{
    k_exact_moe_combine<<<ifa_cdiv(n, 256), 256, 0, s>>>(f, expert_out, weight_dev, (int)n);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int exact_act_mul(int kind, const half_t *a, const half_t *gate, size_t n, half_t *y, hipStream_t s)
{
    IFA_REQUIRE(kind == 0 || kind == 2, "exact activation: kind %d (SiLU and ReLU only: GELU goes through tanhf)", kind);
    IFA_REQUIRE(exact_init(s) == IFA_OK, "exact activation: table upload failed");
    k_exact_act_mul<<<ifa_cdiv(n, 256), 256, 0, s>>>(a, gate, (int)n, kind, y);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

} // namespace ifa

// ---------------------------------------------------------------- the same kernels behind the C ABI, one op at a time: the second link of
// the parity chain (tests/test_gpu_fullsize_oracle.py: the worker's order-exact step == the oracle bit for bit; here every timed op
// against its order-exact form on the SAME inputs, at full size, on the device)
extern "C" {

int ifa_exact_rmsnorm(const void *x_f16, size_t rows, size_t cols, const void *w_f16, const void *b_f16, float multi_base, float eps,
                      void *y_f16, ifa_stream stream)
{
    IFA_REQUIRE(x_f16 && y_f16 && rows > 0 && cols > 0 && rows < (1u << 30) && cols < (1u << 30), "ifa_exact_rmsnorm: bad arguments");
    return ifa::exact_rmsnorm((const ifa::half_t *)x_f16, (int)rows, (int)cols, (const ifa::half_t *)w_f16, (const ifa::half_t *)b_f16, multi_base, eps,
                              (ifa::half_t *)y_f16, ifa_s(stream));
}

int ifa_exact_gemv(int w_dtype, const void *W, size_t rows, size_t cols, int x_dtype, const void *x, const void *bias_f16, void *y_f16, ifa_stream stream)
{
    IFA_REQUIRE(W && x && y_f16 && rows > 0 && cols > 0 && rows < (1u << 30) && cols < (1u << 30), "ifa_exact_gemv: bad arguments");
    if (x_dtype == ifa::Q8_B32T2)
        return ifa::exact_gemv_ax8(w_dtype, W, rows, cols, x, (const ifa::half_t *)bias_f16, (ifa::half_t *)y_f16, ifa_s(stream));
    IFA_REQUIRE(x_dtype == ifa::F16, "ifa_exact_gemv: x dtype %d (F16 or Q8_B32T2)", x_dtype);
    return ifa::exact_gemv_f16x(w_dtype, W, rows, cols, (const ifa::half_t *)x, (const ifa::half_t *)bias_f16, (ifa::half_t *)y_f16, ifa_s(stream));
}

int ifa_exact_attention(const void *q_f16, const void *kcache, const void *vcache, int kv_dtype, int n_ctx, int heads, int kv_heads, int head_dim,
                        float kq_scale, void *out_f16, ifa_stream stream)
{
    IFA_REQUIRE(q_f16 && kcache && vcache && out_f16 && n_ctx > 0 && heads > 0 && kv_heads > 0 && heads % kv_heads == 0 && head_dim > 0,
                "ifa_exact_attention: bad arguments");
    IFA_REQUIRE(kv_dtype == ifa::F16 || (kv_dtype == ifa::Q8_B32T2 && (kv_heads * head_dim) % 32 == 0), "ifa_exact_attention: cache dtype %d", kv_dtype);
    const size_t kvd = (size_t)kv_heads * head_dim;
    const size_t row_bytes = kv_dtype == ifa::Q8_B32T2 ? kvd / 32 * 34 : kvd * 2;
    const float alpha = 1.0f / sqrtf((float)head_dim) / kq_scale;
    return ifa::exact_attention((const ifa::half_t *)q_f16, kcache, vcache, kv_dtype, row_bytes, n_ctx, heads, kv_heads, head_dim, alpha, kq_scale,
                                (ifa::half_t *)out_f16, ifa_s(stream));
}

int ifa_exact_activation_mul(int kind, const void *a_f16, const void *gate_f16, size_t n, void *y_f16, ifa_stream stream)
{
    IFA_REQUIRE(a_f16 && y_f16 && n > 0 && n < (1u << 31), "ifa_exact_activation_mul: bad arguments");
    return ifa::exact_act_mul(kind, (const ifa::half_t *)a_f16, (const ifa::half_t *)gate_f16, n, (ifa::half_t *)y_f16, ifa_s(stream));
}

} // extern "C"

