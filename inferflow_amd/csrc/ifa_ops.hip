// ifa_ops.hip -- op-level counterparts of TensorOpr::{LayerNormalization,
// PositionEmbedding, SoftMax, Activation, Mul, Add, Scale} and the greedy
// argmax, one launch per op, no hidden synchronisation.  These mirror the
// reference op set for drop-in use and for op-by-op parity tests; the decode
// engine (ifa_decode.hip) fuses the same device functions.
#include "ifa_host.h"
#include "ifa_device.h"
#include "ifa_math.h"

namespace ifa {

// Tensor_RmsNorm_Kernel / Tensor_StdNorm_Kernel (src/kernels/unary_tensor_opr.h:216-289, :68-149),
// launcher block (128,1), grid (1,rows) (src/tensor/tensor_opr.cu:511-520, :568-577).
// Std norm keeps the reference's partial-sum structure (128 partials, serial final sum); RMS uses the canonical order
// of ifa_math.h shared with the fused decode prologues (no serial chain; <= 1 half ulp from the reference's order).
template <int KIND>
__global__ void __launch_bounds__(128) k_layernorm(const half_t *__restrict__ x, int cols,
                                                   const half_t *__restrict__ w, const half_t *__restrict__ b,
                                                   float multi_base, float eps, half_t *__restrict__ y,
                                                   const half_t *__restrict__ add = nullptr, half_t *__restrict__ sum_out = nullptr)
{
    __shared__ float part[128];
    __shared__ float part2[128];
    __shared__ float stat[2];
    extern __shared__ __attribute__((aligned(16))) char smem_row[];
    const int tid = threadIdx.x;
    const half_t *src = x + (size_t)blockIdx.x * cols;
    half_t *dst = y + (size_t)blockIdx.x * cols;
    bool staged = false;
    if (add) {      // TensorOpr::Add in front of the norm, same launch: the half-rounded sum goes to sum_out and to LDS
        half_t *row = reinterpret_cast<half_t *>(smem_row);
        const half_t *ad = add + (size_t)blockIdx.x * cols;
        half_t *so = sum_out + (size_t)blockIdx.x * cols;
        if ((cols & 7) == 0) {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            for (int c = tid; c < cols / 8; c += 128) {
                const h8 a8 = reinterpret_cast<const h8 *>(src)[c], b8 = reinterpret_cast<const h8 *>(ad)[c];
                h8 o8;
#pragma unroll
                for (int e = 0; e < 8; e++) o8[e] = f2h((float)a8[e] + (float)b8[e]);
                reinterpret_cast<h8 *>(row)[c] = o8; reinterpret_cast<h8 *>(so)[c] = o8;
            }
        } else {
            for (int xi = tid; xi < cols; xi += 128) {
                const half_t v = f2h(h2f(src[xi]) + h2f(ad[xi]));
                row[xi] = v; so[xi] = v;
            }
        }
        __syncthreads();
        src = row; staged = true;
    }
    if constexpr (KIND == 0) {
        // the canonical order of ifa_math.h (chunk chains, wave butterfly per 64 chunks, ascending groups)
        const int nchunks = (cols + 7) >> 3, ngroups = (nchunks + 63) >> 6;
        const int wv_ = tid >> 6, ln_ = tid & 63;
        for (int g = wv_; g < ngroups; g += 2) {
            const int c = 64 * g + ln_;
            rms_h8 v8;
#pragma unroll
            for (int e = 0; e < 8; e++) v8[e] = (half_t)0;
            if (c < nchunks) {
                const half_t *pc = src + (size_t)c * 8;
                if (c * 8 + 8 <= cols && (reinterpret_cast<uintptr_t>(pc) & 15) == 0) v8 = *reinterpret_cast<const rms_h8 *>(pc);
                else {
#pragma unroll
                    for (int e = 0; e < 8; e++) if (c * 8 + e < cols) v8[e] = pc[e];
                }
            }
            const float pg = wave_sum(rms_chunk_sq(v8));
            if (ln_ == 0) part[g] = pg;
        }
        __syncthreads();
        const float scale = rms_scale_of(rms_total(part, ngroups), cols, eps);
        if ((cols & 7) == 0) {       // element-wise: 16-byte accesses, chunks interleaved over the threads
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            for (int c = tid; c < cols / 8; c += 128) {
                const h8 xv = reinterpret_cast<const h8 *>(src)[c];
                h8 wv, bv, ov;
                if (w) wv = reinterpret_cast<const h8 *>(w)[c];
                if (w && b) bv = reinterpret_cast<const h8 *>(b)[c];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float v = (float)xv[e] * scale;          // rms_apply
                    if (w) {
                        const float mlt = multi_base + (float)wv[e];
                        v = v * mlt;
                        if (b) v = v + (float)bv[e];
                    }
                    ov[e] = f2h(v);
                }
                reinterpret_cast<h8 *>(dst)[c] = ov;
            }
        } else {
            const int x_len = (cols + 127) / 128;
            const int xs0 = tid * x_len, xe = min((tid + 1) * x_len, cols);
            for (int xi = xs0; xi < xe; xi++)
                dst[xi] = f2h(rms_apply(h2f(src[xi]), scale, w ? w + xi : nullptr, b ? b + xi : nullptr, multi_base));
        }
    } else {
        // the row is staged in LDS with wide loads first: the strided per-thread sums below (the reference's order)
        // would otherwise walk global memory one 2-byte load at a time on a latency chain
        half_t *row = reinterpret_cast<half_t *>(smem_row);
        if (!staged) {
            if ((cols & 7) == 0) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                for (int c = tid; c < cols / 8; c += 128) reinterpret_cast<u4 *>(row)[c] = reinterpret_cast<const u4 *>(src)[c];
            } else {
                for (int xi = tid; xi < cols; xi += 128) row[xi] = src[xi];
            }
            __syncthreads();
            src = row;
        }
        float sum = 0.0f, sum2 = 0.0f;
        int xi = tid;
        for (; xi + 7 * 128 < cols; xi += 8 * 128) {     // 8 LDS reads in flight, then the ordered chain
            half_t hv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) hv[u] = src[xi + u * 128];
            // Tensor_StdNorm_Kernel adds in double and rounds to float each step; for half inputs that equals the fp32
            // add / fma (v and v*v are exact in fp32; see rms_partial in ifa_math.h for the rounding argument)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float v = h2f(hv[u]);
                sum = sum + v;
                sum2 = __builtin_fmaf(v, v, sum2);
            }
        }
        for (; xi < cols; xi += 128) {
            const float v = h2f(src[xi]);
            sum = sum + v;
            sum2 = __builtin_fmaf(v, v, sum2);
        }
        part[tid] = sum; part2[tid] = sum2;
        __syncthreads();
        if (tid == 0) {
            float ts = 0.0f, ts2 = 0.0f;
            for (int i0 = 0; i0 < 128; i0 += 32) {           // strictly ordered, LDS reads batched
                float a[32], b2[32];
#pragma unroll
                for (int i = 0; i < 32; i++) { a[i] = part[i0 + i]; b2[i] = part2[i0 + i]; }
#pragma unroll
                for (int i = 0; i < 32; i++) { ts = ts + a[i]; ts2 = ts2 + b2[i]; }
            }
            float mean = ts / (float)cols;
            float mm = mean * mean;
            float var = ts2 / (float)cols - mm;
            stat[0] = mean;
            stat[1] = 1.0f / sqrtf(var + eps);
        }
        __syncthreads();
        const float mean = stat[0], scale = stat[1];
        if ((cols & 7) == 0) {       // element-wise: 8 contiguous elements per thread and step, 16-byte accesses
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            for (int c = tid; c < cols / 8; c += 128) {
                const h8 xv = reinterpret_cast<const h8 *>(src)[c];
                h8 wv, bv, ov;
                if (w) wv = reinterpret_cast<const h8 *>(w)[c];
                if (w && b) bv = reinterpret_cast<const h8 *>(b)[c];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float v = ((float)xv[e] - mean) * scale;
                    if (w) v = v * (float)wv[e];
                    if (w && b) v = v + (float)bv[e];
                    ov[e] = f2h(v);
                }
                reinterpret_cast<h8 *>(dst)[c] = ov;
            }
        } else {
            for (int xi = tid; xi < cols; xi += 128) {
                float v = (h2f(src[xi]) - mean) * scale;
                if (w) v = v * h2f(w[xi]);
                if (w && b) v = v + h2f(b[xi]);
                dst[xi] = f2h(v);
            }
        }
    }
}

// PosEmbedding_Rope_Order2_Kernel / _Std_Kernel (unary_tensor_opr.h:661-740)
// One thread per (token, rotated pair): the angle (two powf, cosf, sinf -- the dominant cost of the per-element
// form) is evaluated once and applied to that pair of every head with rope_rotate's expressions.
__global__ void __launch_bounds__(256) k_rope(half_t *__restrict__ x, int head_dim, int heads, int tokens, int pos0,
                                              float theta, int order, int rope_dims, int rope_cols,
                                              const int *__restrict__ pos_tab = nullptr)
{
    const int half_dim = head_dim / 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)tokens * half_dim) return;
    const int col = (int)(idx % half_dim);
    const int t = (int)(idx / half_dim);
    if (order == 2 && 2 * col >= rope_cols) return;
    float c, s;
    rope_angle(col, pos_tab ? pos_tab[t] : pos0 + t, theta, rope_dims, c, s);      // pos_tab: one position per row
    const int i0 = order == 2 ? col : 2 * col, i1 = order == 2 ? col + rope_cols / 2 : 2 * col + 1;
    // blockIdx.y splits the heads when there are too few (token, pair) threads to fill the chip (short prompts)
    const int hpb = (heads + (int)gridDim.y - 1) / (int)gridDim.y, h0 = (int)blockIdx.y * hpb, h1 = min(heads, h0 + hpb);
    half_t *row = x + ((size_t)t * heads + h0) * head_dim;
    for (int h = h0; h < h1; h++, row += head_dim) {
        const float x0 = h2f(row[i0]), x1 = h2f(row[i1]);
        float a = x0 * c, bq = x1 * s, d = x0 * s, e = x1 * c;
        row[i0] = f2h(a - bq);
        row[i1] = f2h(d + e);
    }
}

// PositionEmbedding(q), PositionEmbedding(k), SetKRows, SetVRows of a prefill step in ONE launch, a workgroup per token:
// the token's head_dim/2 angles are evaluated once (LDS), q is rotated in place, k is rotated through LDS and leaves as
// 16-byte stores to the k buffer and -- F16 caches -- to its cache row, v is copied to its cache row.
__global__ void __launch_bounds__(256) k_rope_qk_store(half_t *__restrict__ q, half_t *__restrict__ k, const half_t *__restrict__ v,
                                                       int head_dim, int heads, int kv_heads, int pos0, float theta, int order,
                                                       int rope_dims, int rope_cols, half_t *__restrict__ kcache,
                                                       half_t *__restrict__ vcache, size_t cache_row_elems)
{
    extern __shared__ __attribute__((aligned(16))) char smem_rk[];
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const int t = blockIdx.x, tid = threadIdx.x, half_dim = head_dim / 2;
    float *cs = reinterpret_cast<float *>(smem_rk);                       // [half_dim][2]
    half_t *krow = reinterpret_cast<half_t *>(cs + 2 * half_dim);         // [kv_heads * head_dim]
    const int kv_dim = kv_heads * head_dim;
    for (int c = tid; c < half_dim; c += 256) {
        float co = 1.0f, si = 0.0f;
        if (order != 2 || 2 * c < rope_cols) rope_angle(c, pos0 + t, theta, rope_dims, co, si);
        cs[2 * c] = co; cs[2 * c + 1] = si;
    }
    const half_t *ksrc = k + (size_t)t * kv_dim;
    for (int c = tid; c < kv_dim / 8; c += 256) reinterpret_cast<u4 *>(krow)[c] = reinterpret_cast<const u4 *>(ksrc)[c];
    if (vcache)
        for (int c = tid; c < kv_dim / 8; c += 256)
            reinterpret_cast<u4 *>(vcache + (size_t)t * cache_row_elems)[c] = reinterpret_cast<const u4 *>(v + (size_t)t * kv_dim)[c];
    __syncthreads();
    auto rotate = [&](half_t *row, int col) {                              // rope_rotate's expressions (ifa_math.h)
        if (order == 2 && 2 * col >= rope_cols) return;
        const int i0 = order == 2 ? col : 2 * col, i1 = order == 2 ? col + rope_cols / 2 : 2 * col + 1;
        const float c = cs[2 * col], sn = cs[2 * col + 1];
        const float x0 = h2f(row[i0]), x1 = h2f(row[i1]);
        float a = x0 * c, bq = x1 * sn, d = x0 * sn, e = x1 * c;
        row[i0] = f2h(a - bq);
        row[i1] = f2h(d + e);
    };
    half_t *qrow = q + (size_t)t * heads * head_dim;
    if (rope_cols == head_dim && head_dim % 16 == 0) {
        // eight pairs per thread through 16-byte accesses (the same expressions as `rotate`): a 128-token prompt spent 10 us per layer
        // here in 2-byte loads and stores (rocprofv3 r06), now one round trip
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const int per_head = half_dim / 8;
        for (int p = tid; p < heads * per_head; p += 256) {
            const int h = p / per_head, c0 = (p % per_head) * 8;
            half_t *row = qrow + (size_t)h * head_dim;
            h8 *pa = reinterpret_cast<h8 *>(row + (order == 2 ? c0 : 2 * c0)), *pb = reinterpret_cast<h8 *>(row + (order == 2 ? c0 + half_dim : 2 * c0 + 8));
            const h8 va = *pa, vb = *pb;
            h8 oa, ob;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                // order 2: pair e = (va[e], vb[e]); adjacent pairs: pair e = elements 2e, 2e + 1 of the 16 halves va | vb
                const float x0 = order == 2 ? h2f(va[e]) : h2f(e < 4 ? va[2 * e] : vb[2 * e - 8]);
                const float x1 = order == 2 ? h2f(vb[e]) : h2f(e < 4 ? va[2 * e + 1] : vb[2 * e - 7]);
                const float c = cs[2 * (c0 + e)], sn = cs[2 * (c0 + e) + 1];
                const float a = x0 * c, bq = x1 * sn, d = x0 * sn, ee = x1 * c;
                const half_t r0 = f2h(a - bq), r1 = f2h(d + ee);
                if (order == 2) { oa[e] = r0; ob[e] = r1; }
                else if (e < 4) { oa[2 * e] = r0; oa[2 * e + 1] = r1; }
                else { ob[2 * e - 8] = r0; ob[2 * e - 7] = r1; }
            }
            *pa = oa; *pb = ob;
        }
    } else
    for (int p = tid; p < heads * half_dim; p += 256) rotate(qrow + (size_t)(p / half_dim) * head_dim, p % half_dim);
    for (int p = tid; p < kv_heads * half_dim; p += 256) rotate(krow + (size_t)(p / half_dim) * head_dim, p % half_dim);
    __syncthreads();
    for (int c = tid; c < kv_dim / 8; c += 256) {
        const u4 kv8 = reinterpret_cast<const u4 *>(krow)[c];
        reinterpret_cast<u4 *>(k + (size_t)t * kv_dim)[c] = kv8;
        if (kcache) reinterpret_cast<u4 *>(kcache + (size_t)t * cache_row_elems)[c] = kv8;
    }
}

// PosEmbedding_Alibi_Std_Kernel (unary_tensor_opr.h:742-762)
__global__ void __launch_bounds__(256) k_alibi(half_t *__restrict__ s, int ctx, int q_tokens, int heads,
                                               int base_head, int total_heads)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)heads * q_tokens * ctx;
    if (idx >= total) return;
    const int col = (int)(idx % ctx);
    const int h = (int)(idx / ((size_t)ctx * q_tokens));
    const float mk = alibi_slope(h + base_head, total_heads);
    float a = (float)col * mk;
    s[idx] = f2h(a + h2f(s[idx]));
}

// Tensor_SoftMax_Alg2_Kernel (unary_tensor_opr.h:480-535): 32 lanes per row
// (CUDA warp) with strided partial sums and an xor butterfly; reproduced on a
// half-wave so the summation order matches the restated reference.
__global__ void __launch_bounds__(64) k_softmax(half_t *__restrict__ s, int cx, int cy, size_t nrows, int prefix_len,
                                               float scale)
{
    const int lane32 = threadIdx.x & 31;
    const int sub = threadIdx.x >> 5;
    const size_t rowi = (size_t)blockIdx.x * 2 + sub;
    if (rowi >= nrows) return;
    half_t *row = s + rowi * cx;
    const int r = (int)(rowi % cy);
    float mx = -INFINITY;
    for (int xi = lane32; xi < cx; xi += 32) {
        float v = scale * h2f(row[xi]);
        if (prefix_len >= 0 && xi > prefix_len + r) v = -INFINITY;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 32));
    float sum = 0.0f;
    for (int xi = lane32; xi < cx; xi += 32) {
        float v = scale * h2f(row[xi]);
        if (prefix_len >= 0 && xi > prefix_len + r) v = -INFINITY;
        const float e = expf(v - mx);
        sum = sum + e;
        row[xi] = f2h(e);
    }
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) sum = sum + __shfl_xor(sum, m, 32);
    const float inv = 1.0f / sum;
    for (int xi = lane32; xi < cx; xi += 32) row[xi] = f2h(h2f(row[xi]) * inv);
}

__global__ void __launch_bounds__(256) k_activation(const half_t *__restrict__ x, size_t rows, size_t cols, int kind,
                                                    int is_glu, half_t *__restrict__ y)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const size_t r = idx / cols, c = idx % cols;
    const size_t in_off = is_glu ? (r * 2 * cols + c) : idx;
    float fx = act_fn(h2f(x[in_off]), kind);
    if (is_glu) fx = fx * h2f(x[in_off + cols]);
    y[idx] = f2h(fx);
}

// Activation followed by Mul in one pass (8 elements per thread), each with its own half rounding like the two ops
__global__ void __launch_bounds__(256) k_act_mul(const half_t *__restrict__ a, const half_t *__restrict__ b, size_t n8, int kind,
                                                 half_t *__restrict__ c)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const u32x4 ra = reinterpret_cast<const u32x4 *>(a)[i], rb = reinterpret_cast<const u32x4 *>(b)[i];
    const half_t *ha = reinterpret_cast<const half_t *>(&ra), *hb = reinterpret_cast<const half_t *>(&rb);
    half_t o[8];
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2h(h2f(f2h(act_fn(h2f(ha[e]), kind))) * h2f(hb[e]));
    reinterpret_cast<u32x4 *>(c)[i] = *reinterpret_cast<const u32x4 *>(o);
}

__global__ void __launch_bounds__(256) k_mul(const half_t *a, const half_t *b, size_t n, half_t *c)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = f2h(h2f(a[i]) * h2f(b[i]));
}
__global__ void __launch_bounds__(256) k_add(const half_t *a, const half_t *b, size_t n, size_t period, half_t *c)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = f2h(h2f(a[i]) + h2f(b[period ? i % period : i]));
}
__global__ void __launch_bounds__(256) k_scale(const half_t *a, float s, size_t n, half_t *c)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = f2h(h2f(a[i]) * s);
}

// greedy top-1: first maximum wins (GetSortedTopK, sampling_strategy.cc:372-386)
// excl: optional {count (<= 3), id, id, id}: ids the queue is never offered -- the vocabulary's unk id and Invalid-type
// tokens (sampling_strategy.cc:281-297)
// (a workgroup per row: blockIdx.x-th row at v + blockIdx.x * row_stride, result at out[blockIdx.x])
__global__ void __launch_bounds__(1024) k_argmax(const half_t *__restrict__ v, size_t n, int *__restrict__ out, const int *__restrict__ excl,
                                                 size_t row_stride = 0)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    v += (size_t)blockIdx.x * row_stride; out += blockIdx.x;
    const int ne = excl ? min(max(excl[0], 0), 3) : 0;
    const int e0 = ne > 0 ? excl[1] : -1, e1 = ne > 1 ? excl[2] : -1, e2 = ne > 2 ? excl[3] : -1;
    float best = -INFINITY; int besti = 0x7FFFFFFF;
    argmax_scan(v, n, e0, e1, e2, (int)threadIdx.x, (int)blockDim.x, best, besti);
    if (besti == 0x7FFFFFFF) { best = -INFINITY; }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        float ob = __shfl_xor(best, m); int oi = __shfl_xor(besti, m);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        *out = besti == 0x7FFFFFFF ? 0 : besti;
    }
}

// B[idx[r]][:] = hfma(A[r][:], w[r], B[idx[r]][:])   (AddByRowIdx_Kernel, src/kernels/binary_tensor_opr.h:80-125)
__global__ void __launch_bounds__(256) k_add_by_row_index(half_t *__restrict__ B, const half_t *__restrict__ A, int rows, int cols,
                                                          const int *__restrict__ idx, const half_t *__restrict__ w)
{
    const int r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows || c >= cols) return;
    const half_t wr = w ? w[r] : (half_t)1.0f;
    half_t *b = B + (size_t)idx[r] * cols + c;
    *b = __builtin_fmaf16(A[(size_t)r * cols + c], wr, *b);
}

// HostTensorOpr::BuildRowsForMoE's per-row part (src/tensor/host_tensor_opr.cc:190-244) on the device, one thread per
// token row: top-k by repeated first-maximum over the softmaxed router row, probabilities below 1e-5 dropped, optional
// renormalisation over the kept ones; the kept (expert, weight) pairs in ASCENDING expert id (the order the reference
// visits experts in), unused slots expert -1 / weight 0.
// One wave per row: lane e holds expert e's probability (E <= 64).
__global__ void __launch_bounds__(256) k_moe_route_rows(const half_t *__restrict__ probs_h, int T, int E, int top_k, int norm,
                                                        int *__restrict__ sel, half_t *__restrict__ wout)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float p = lane < E ? h2f(probs_h[(size_t)t * E + lane]) : -INFINITY;
    bool used = lane >= E;
    int idx[8]; float w[8];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        idx[k] = 0x7FFFFFFF; w[k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= top_k || k >= E) break;
        const float mx = wave_max(used ? -INFINITY : p);
        const unsigned long long cand = __ballot(!used && p == mx);       // first maximum wins, like the host sort
        if (!cand) break;
        const int best = __ffsll((long long)cand) - 1;
        if (lane == best) used = true;
        const float pb = __shfl(p, best);
        if (pb < 0.00001f) continue;
        // (n is wave-uniform: a static slot per k would leave holes, so the kept pairs are compacted with selects)
#pragma unroll
        for (int j = 0; j < 8; j++) if (j == n) { idx[j] = best; w[j] = pb; }
        n++;
    }
    if (norm && n > 0) {
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) if (j < n) sum = sum + w[j];
#pragma unroll
        for (int j = 0; j < 8; j++) if (j < n) w[j] = w[j] / sum;
    }
    if (lane != 0) return;
    // kept pairs in ascending expert id: slot of pair j = number of kept experts below it
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j >= n) continue;
        int rank = 0;
#pragma unroll
        for (int j2 = 0; j2 < 8; j2++) rank += (j2 < n && idx[j2] < idx[j]) ? 1 : 0;
        sel[(size_t)t * top_k + rank] = idx[j];
        wout[(size_t)t * top_k + rank] = f2h(w[j]);
    }
    for (int slot = n; slot < top_k; slot++) { sel[(size_t)t * top_k + slot] = -1; wout[(size_t)t * top_k + slot] = (half_t)0; }
}

} // namespace ifa

using namespace ifa;

extern "C" {

// engine-internal: sum_out = a + addend (TensorOpr::Add, half rounding), y = LayerNormalization(sum_out) in one launch
int ifa_add_layernorm(int kind, const void *a, const void *addend, size_t rows, size_t cols, const void *w, const void *b,
                      float multi_base, float eps, void *sum_out, void *y, ifa_stream stream)
{
    IFA_REQUIRE(a && addend && sum_out && y, "ifa_add_layernorm: null pointer");
    IFA_REQUIRE(kind == 0 || kind == 1, "ifa_add_layernorm: kind %d", kind);
    IFA_REQUIRE(b == nullptr || w != nullptr, "ifa_add_layernorm: bias without weight");
    if (rows == 0 || cols == 0) return IFA_OK;
    const size_t smem = (cols * 2 + 15) & ~(size_t)15;
    IFA_REQUIRE(smem <= 64 * 1024, "ifa_add_layernorm: %zu columns", cols);
    if (kind == 0)
        k_layernorm<0><<<dim3((unsigned)rows), dim3(128), smem, ifa_s(stream)>>>((const half_t *)a, (int)cols, (const half_t *)w, (const half_t *)b, multi_base, eps,
                                                                                 (half_t *)y, (const half_t *)addend, (half_t *)sum_out);
    else
        k_layernorm<1><<<dim3((unsigned)rows), dim3(128), smem, ifa_s(stream)>>>((const half_t *)a, (int)cols, (const half_t *)w, (const half_t *)b, multi_base, eps,
                                                                                 (half_t *)y, (const half_t *)addend, (half_t *)sum_out);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_layernorm(int kind, const void *x, size_t rows, size_t cols, const void *w, const void *b,
                  float multi_base, float eps, void *y, ifa_stream stream)
{
    IFA_REQUIRE(x && y, "ifa_layernorm: null pointer");
    IFA_REQUIRE(kind == 0 || kind == 1, "ifa_layernorm: kind %d", kind);
    IFA_REQUIRE(b == nullptr || w != nullptr, "ifa_layernorm: bias without weight");
    if (rows == 0 || cols == 0) return IFA_OK;
    IFA_REQUIRE(cols <= 65536, "ifa_layernorm: %zu columns (limit 65536)", cols);
    if (kind == 0)
        k_layernorm<0><<<dim3((unsigned)rows), dim3(128), 0, ifa_s(stream)>>>((const half_t *)x, (int)cols, (const half_t *)w, (const half_t *)b, multi_base, eps, (half_t *)y);
    else
        k_layernorm<1><<<dim3((unsigned)rows), dim3(128), (cols * 2 + 15) & ~(size_t)15, ifa_s(stream)>>>((const half_t *)x, (int)cols, (const half_t *)w, (const half_t *)b, multi_base, eps, (half_t *)y);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

static unsigned rope_head_split(size_t token_pairs, int heads)
{
    // aim at >= 32K threads in flight; one head per thread at most
    size_t want = (32768 + token_pairs - 1) / token_pairs;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(want, (size_t)heads));
}

int ifa_rope(void *x, int head_dim, int heads, int tokens, int pos0, float theta, int order,
             float partial_rotary_factor, ifa_stream stream)
{
    IFA_REQUIRE(x, "ifa_rope: null pointer");
    IFA_REQUIRE(order == 1 || order == 2, "ifa_rope: order %d", order);
    IFA_REQUIRE(head_dim > 0 && head_dim % 2 == 0, "ifa_rope: head_dim %d", head_dim);
    if (tokens <= 0 || heads <= 0) return IFA_OK;
    if (partial_rotary_factor <= 0) partial_rotary_factor = 1.0f;
    // src/tensor/tensor_opr.cu:701-702 (F16 path passes rope_dims, appendix A12)
    int rope_cols = (int)(head_dim * partial_rotary_factor + 0.5f);
    int rope_dims = rope_cols;
    size_t total = (size_t)tokens * (head_dim / 2);
    k_rope<<<dim3(ifa_cdiv(total, 256), rope_head_split(total, heads)), dim3(256), 0, ifa_s(stream)>>>((half_t *)x, head_dim, heads, tokens, pos0, theta, order, rope_dims, rope_cols);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// engine-internal: RoPE of q and k (+ the F16 cache rows of k and v when kcache_rows / vcache_rows are given: the rows of
// positions pos0 .. pos0 + tokens - 1, cache_row_elems halfs apart) in one launch; IFA_ERR_STATE when the shape is not covered
int ifa_rope_qk_store(void *q, void *k, const void *v, int head_dim, int heads, int kv_heads, int tokens, int pos0, float theta,
                      int order, float partial_rotary_factor, void *kcache_rows, void *vcache_rows, size_t cache_row_elems,
                      ifa_stream stream)
{
    IFA_REQUIRE(q && k, "ifa_rope_qk_store: null pointer");
    if (tokens <= 0) return IFA_OK;
    const int kv_dim = kv_heads * head_dim;
    if ((order != 1 && order != 2) || head_dim % 2 || kv_dim % 8 || (vcache_rows && !v)) return IFA_ERR_STATE;
    if (partial_rotary_factor <= 0) partial_rotary_factor = 1.0f;
    const int rope_cols = (int)(head_dim * partial_rotary_factor + 0.5f);
    const size_t smem = (size_t)head_dim * 4 + (size_t)kv_dim * 2;
    if (smem > 64 * 1024 || (((uintptr_t)k | (uintptr_t)kcache_rows | (uintptr_t)vcache_rows | (uintptr_t)v) & 15) || (cache_row_elems % 8)) return IFA_ERR_STATE;
    k_rope_qk_store<<<dim3((unsigned)tokens), dim3(256), smem, ifa_s(stream)>>>((half_t *)q, (half_t *)k, (const half_t *)v, head_dim, heads, kv_heads, pos0,
                                                                               theta, order, rope_cols, rope_cols, (half_t *)kcache_rows,
                                                                               (half_t *)vcache_rows, cache_row_elems);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_rope_rows(void *x, int head_dim, int heads, int tokens, const int *positions_dev, float theta, int order,
                  float partial_rotary_factor, ifa_stream stream)
{
    IFA_REQUIRE(x && positions_dev, "ifa_rope_rows: null pointer");
    IFA_REQUIRE(order == 1 || order == 2, "ifa_rope_rows: order %d", order);
    IFA_REQUIRE(head_dim > 0 && head_dim % 2 == 0, "ifa_rope_rows: head_dim %d", head_dim);
    if (tokens <= 0 || heads <= 0) return IFA_OK;
    if (partial_rotary_factor <= 0) partial_rotary_factor = 1.0f;
    int rope_cols = (int)(head_dim * partial_rotary_factor + 0.5f);
    size_t total = (size_t)tokens * (head_dim / 2);
    k_rope<<<dim3(ifa_cdiv(total, 256), rope_head_split(total, heads)), dim3(256), 0, ifa_s(stream)>>>((half_t *)x, head_dim, heads, tokens, 0, theta, order, rope_cols,
                                                                       rope_cols, positions_dev);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_alibi(void *scores, int ctx, int q_tokens, int heads, int base_head, int total_heads, ifa_stream stream)
{
    IFA_REQUIRE(scores, "ifa_alibi: null pointer");
    size_t total = (size_t)heads * q_tokens * ctx;
    if (total == 0) return IFA_OK;
    k_alibi<<<dim3(ifa_cdiv(total, 256)), dim3(256), 0, ifa_s(stream)>>>((half_t *)scores, ctx, q_tokens, heads, base_head, total_heads);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_softmax(void *s, int cx, int cy, int cz, int prefix_len, float scale, ifa_stream stream)
{
    IFA_REQUIRE(s, "ifa_softmax: null pointer");
    size_t nrows = (size_t)cy * cz;
    if (nrows == 0 || cx <= 0) return IFA_OK;
    IFA_REQUIRE(nrows <= 0x7FFFFFFFu, "ifa_softmax: too many rows");
    k_softmax<<<dim3(ifa_cdiv(nrows, 2)), dim3(64), 0, ifa_s(stream)>>>((half_t *)s, cx, cy, nrows, prefix_len, scale);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_activation(int kind, int is_glu, const void *x, size_t rows, size_t cols, void *y, ifa_stream stream)
{
    IFA_REQUIRE(x && y, "ifa_activation: null pointer");
    IFA_REQUIRE(kind >= 0 && kind <= 2, "ifa_activation: kind %d", kind);
    if (rows * cols == 0) return IFA_OK;
    k_activation<<<dim3(ifa_cdiv(rows * cols, 256)), dim3(256), 0, ifa_s(stream)>>>((const half_t *)x, rows, cols, kind, is_glu, (half_t *)y);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// engine-internal: c = Mul(Activation(a), b) -- TensorOpr::Activation + TensorOpr::Mul of the gated FFN in one launch
int ifa_activation_mul(int kind, const void *a, const void *b, size_t n, void *c, ifa_stream stream)
{
    IFA_REQUIRE(a && b && c, "ifa_activation_mul: null pointer");
    IFA_REQUIRE(kind >= 0 && kind <= 2, "ifa_activation_mul: kind %d", kind);
    if (n == 0) return IFA_OK;
    if (n % 8 || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15)) {
        int rc = ifa_activation(kind, 0, a, 1, n, c, stream);
        return rc ? rc : ifa_mul(c, b, n, c, stream);
    }
    k_act_mul<<<dim3(ifa_cdiv(n / 8, 256)), dim3(256), 0, ifa_s(stream)>>>((const half_t *)a, (const half_t *)b, n / 8, kind, (half_t *)c);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_mul(const void *a, const void *b, size_t n, void *c, ifa_stream stream)
{
    IFA_REQUIRE(a && b && c, "ifa_mul: null pointer");
    if (n == 0) return IFA_OK;
    k_mul<<<dim3(ifa_cdiv(n, 256)), dim3(256), 0, ifa_s(stream)>>>((const half_t *)a, (const half_t *)b, n, (half_t *)c);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_add(const void *a, const void *b, size_t n, size_t b_period, void *c, ifa_stream stream)
{
    IFA_REQUIRE(a && b && c, "ifa_add: null pointer");
    if (n == 0) return IFA_OK;
    k_add<<<dim3(ifa_cdiv(n, 256)), dim3(256), 0, ifa_s(stream)>>>((const half_t *)a, (const half_t *)b, n, b_period, (half_t *)c);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_scale(const void *a, float s, size_t n, void *c, ifa_stream stream)
{
    IFA_REQUIRE(a && c, "ifa_scale: null pointer");
    if (n == 0) return IFA_OK;
    k_scale<<<dim3(ifa_cdiv(n, 256)), dim3(256), 0, ifa_s(stream)>>>((const half_t *)a, s, n, (half_t *)c);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_argmax(const void *logits, size_t n, int *out_index_dev, ifa_stream stream)
{
    IFA_REQUIRE(logits && out_index_dev, "ifa_argmax: null pointer");
    IFA_REQUIRE(n > 0 && n < 0x7FFFFFFFu, "ifa_argmax: n %zu", n);
    k_argmax<<<dim3(1), dim3(1024), 0, ifa_s(stream)>>>((const half_t *)logits, n, out_index_dev, nullptr);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_argmax_masked(const void *logits, size_t n, const int *excluded_dev, int *out_index_dev, ifa_stream stream)
{
    IFA_REQUIRE(logits && out_index_dev, "ifa_argmax_masked: null pointer");
    IFA_REQUIRE(n > 0 && n < 0x7FFFFFFFu, "ifa_argmax_masked: n %zu", n);
    k_argmax<<<dim3(1), dim3(1024), 0, ifa_s(stream)>>>((const half_t *)logits, n, out_index_dev, excluded_dev);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_moe_route_topk(const void *probs_f16, size_t tokens, int experts, int top_k, int norm_top_k_prob, int *sel_out_dev,
                       void *weights_out_f16_dev, ifa_stream stream)
{
    IFA_REQUIRE(probs_f16 && sel_out_dev && weights_out_f16_dev, "ifa_moe_route_topk: null pointer");
    IFA_REQUIRE(experts >= 1 && experts <= 64 && top_k >= 1 && top_k <= 8, "ifa_moe_route_topk: experts %d / top_k %d out of range", experts, top_k);
    if (tokens == 0) return IFA_OK;
    k_moe_route_rows<<<dim3(ifa_cdiv(tokens, 4)), dim3(256), 0, ifa_s(stream)>>>((const half_t *)probs_f16, (int)tokens, experts, top_k, norm_top_k_prob,
                                                                                  sel_out_dev, (half_t *)weights_out_f16_dev);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// engine-internal: greedy top-1 of `rows` logit rows in one launch (out[r] = argmax of row r)
int ifa_argmax_rows(const void *logits, size_t n, size_t row_stride, size_t rows, int *out_dev, const int *excluded_dev, ifa_stream stream)
{
    IFA_REQUIRE(logits && out_dev && n > 0, "ifa_argmax_rows: bad arguments");
    if (rows == 0) return IFA_OK;
    k_argmax<<<dim3((unsigned)rows), dim3(1024), 0, ifa_s(stream)>>>((const half_t *)logits, n, out_dev, excluded_dev, row_stride);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_kv_store(int kv_dtype, const void *rows_f16, size_t tokens, size_t kv_dim, void *cache, size_t first_row, ifa_stream stream)
{
    IFA_REQUIRE(rows_f16 && cache, "ifa_kv_store: null pointer");
    IFA_REQUIRE(kv_dtype == F16 || kv_dtype == Q8_B32T2, "ifa_kv_store: cache dtype %d (F16 or Q8_B32T2)", kv_dtype);
    if (tokens == 0 || kv_dim == 0) return IFA_OK;
    if (kv_dtype == F16) {       // SetKRows / SetVRows, same-type branch: a row copy (kv_cache.cc:159-201)
        IFA_HIP_CHECK(hipMemcpyAsync((char *)cache + first_row * kv_dim * 2, rows_f16, tokens * kv_dim * 2, hipMemcpyDeviceToDevice, ifa_s(stream)));
        return IFA_OK;
    }
    IFA_REQUIRE(kv_dim % 32 == 0, "ifa_kv_store: Q8 rows need kv_dim %% 32 == 0 (got %zu)", kv_dim);
    // quantising branch: TensorOpr::Quantize -> the Alg2 kernel (kv_cache.cc:203-249)
    return ifa_quantize_act_q8(rows_f16, tokens, kv_dim, (char *)cache + first_row * ifa_row_bytes(Q8_B32T2, kv_dim), stream);
}

int ifa_add_by_row_index(void *b_f16, const void *a_f16, size_t rows, size_t cols, const int *row_idx_dev,
                         const void *weights_f16_dev, ifa_stream stream)
{
    IFA_REQUIRE(b_f16 && a_f16 && row_idx_dev, "ifa_add_by_row_index: null pointer");
    if (rows == 0 || cols == 0) return IFA_OK;
    IFA_REQUIRE(rows < 65536 && cols < (1u << 30), "ifa_add_by_row_index: shape too large");
    k_add_by_row_index<<<dim3(ifa_cdiv(cols, 256), (unsigned)rows), dim3(256), 0, ifa_s(stream)>>>(
        (half_t *)b_f16, (const half_t *)a_f16, (int)rows, (int)cols, row_idx_dev, (const half_t *)weights_f16_dev);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

} // extern "C"
