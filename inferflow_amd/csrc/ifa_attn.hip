// ifa_attn.hip -- attention over a KV cache (op-level, any q_tokens).
//
// Replaces the reference sequence GetKRows(+dequant) / TransposeYZ / RepeatKV /
// Gemm_Alg2 / ALiBi / SoftMax / GetVRows / Transpose / RepeatKV / Gemm_Alg2 /
// TransposeYZ+Assign (src/transformer/inference_worker.cc:1116-1312, :1639-1724)
// with one kernel per (head, query token): GQA by indexing, Q8 rows read in
// place, scores kept in LDS.  Rounding points are the reference's:
//   S = half(alpha * sum_d q.k)        (Gemm_Alg2_Kernel, src/kernels/gemm.h:83-178)
//   P = half(half(exp(scale*S - max)) * (1/sum))   (Tensor_SoftMax_Alg2_Kernel)
//   O = half(sum_j P_j * V_j)          (Gemm_Alg2_Kernel)
// q.k and P.V are accumulated in fp32 in index order (products of halfs are
// exact in fp32), so S and O match the restated reference bit-for-bit given the
// same P; P differs by the device expf and the softmax summation order.
#include "ifa_host.h"
#include "ifa_device.h"
#include "ifa_math.h"

namespace ifa {

// element d of kv-row j: F16 cache or Q8_B32T2 rows (dequantised to half as q*scale)
template <bool Q8>
__device__ __forceinline__ float kv_elem(const uint8_t *cache, size_t row_bytes, int j, int e)
{
    if constexpr (Q8) {
        const uint8_t *blk = cache + (size_t)j * row_bytes + (size_t)(e >> 5) * 34;
        const float scale = hbits2f(*reinterpret_cast<const uint16_t *>(blk));
        const int q = (int)(int8_t)blk[2 + (e & 31)];
        return h2f(f2h((float)q * scale));
    } else {
        return h2f(reinterpret_cast<const half_t *>(cache + (size_t)j * row_bytes)[e]);
    }
}

template <bool Q8>
__global__ void __launch_bounds__(256) k_attention(const half_t *__restrict__ q, const uint8_t *__restrict__ kc,
                                                   const uint8_t *__restrict__ vc, int n_ctx, int q_tokens,
                                                   int prefix_len, int heads, int kv_heads, int head_dim,
                                                   float kq_scale, int alibi, int alibi_base, int alibi_total,
                                                   half_t *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *S = reinterpret_cast<half_t *>(smem);                       // [n_ctx]
    float *red = reinterpret_cast<float *>(smem + (((size_t)n_ctx * 2 + 15) & ~(size_t)15));  // [8]
    const int h = blockIdx.x, t = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kvh = h / (heads / kv_heads);
    const int kv_dim = kv_heads * head_dim;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const half_t *qv = q + ((size_t)t * heads + h) * head_dim;
    const float alpha = 1.0f / sqrtf((float)head_dim) / kq_scale;
    const int n_valid = min(n_ctx, prefix_len + t + 1);   // causal: xi <= prefix_len + t
    const float mk = alibi ? alibi_slope(h + alibi_base, alibi_total) : 0.0f;

    // ---- scores
    float lmax = -INFINITY;
    for (int j = tid; j < n_ctx; j += 256) {
        float c = 0.0f;
        for (int d = 0; d < head_dim; d++) c = __builtin_fmaf(h2f(qv[d]), kv_elem<Q8>(kc, row_bytes, j, kvh * head_dim + d), c);
        half_t s = f2h(alpha * c);
        if (alibi) { float a = (float)j * mk; s = f2h(a + h2f(s)); }
        S[j] = s;
        if (j < n_valid) lmax = fmaxf(lmax, kq_scale * h2f(s));
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    // ---- exp + sum
    float lsum = 0.0f;
    for (int j = tid; j < n_ctx; j += 256) {
        float e = 0.0f;
        if (j < n_valid) e = expf(kq_scale * h2f(S[j]) - mx);
        lsum += e;
        S[j] = f2h(e);
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / (((red[4] + red[5]) + red[6]) + red[7]);
    for (int j = tid; j < n_ctx; j += 256) S[j] = f2h(h2f(S[j]) * inv);
    __syncthreads();
    // ---- O = P.V, one thread per output dim, j ascending (reference order)
    for (int d = tid; d < head_dim; d += 256) {
        float c = 0.0f;
        for (int j = 0; j < n_valid; j++)
            c = __builtin_fmaf(h2f(S[j]), kv_elem<Q8>(vc, row_bytes, j, kvh * head_dim + d), c);
        out[(size_t)t * heads * head_dim + (size_t)h * head_dim + d] = f2h(c);
    }
}

} // namespace ifa

using namespace ifa;

extern "C" int ifa_attention(const void *q, const void *kcache, const void *vcache, int kv_dtype, int n_ctx,
                             int q_tokens, int prefix_len, int heads, int kv_heads, int head_dim, float kq_scale,
                             int alibi, int alibi_base_head, int alibi_total_heads, void *out, ifa_stream stream)
{
    IFA_REQUIRE(q && kcache && vcache && out, "ifa_attention: null pointer");
    IFA_REQUIRE(kv_dtype == F16 || kv_dtype == Q8_B32T2, "ifa_attention: kv dtype %d", kv_dtype);
    IFA_REQUIRE(heads > 0 && kv_heads > 0 && heads % kv_heads == 0, "ifa_attention: heads %d kv_heads %d", heads, kv_heads);
    IFA_REQUIRE(head_dim > 0 && (kv_heads * head_dim) % 32 == 0, "ifa_attention: head_dim %d", head_dim);
    IFA_REQUIRE(n_ctx > 0 && q_tokens > 0 && prefix_len >= 0, "ifa_attention: n_ctx %d q_tokens %d prefix %d", n_ctx, q_tokens, prefix_len);
    IFA_REQUIRE(n_ctx <= 65536 && q_tokens <= 65535, "ifa_attention: context too long for the op-level kernel");
    IFA_REQUIRE(kq_scale > 0, "ifa_attention: kq_scale must be > 0");
    size_t smem = (((size_t)n_ctx * 2 + 15) & ~(size_t)15) + 64;
    dim3 grid((unsigned)heads, (unsigned)q_tokens);
    if (kv_dtype == Q8_B32T2) {
        if (smem > 48 * 1024)
            IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_attention<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        k_attention<true><<<grid, dim3(256), smem, ifa_s(stream)>>>((const half_t *)q, (const uint8_t *)kcache, (const uint8_t *)vcache, n_ctx, q_tokens, prefix_len, heads, kv_heads, head_dim, kq_scale, alibi, alibi_base_head, alibi_total_heads > 0 ? alibi_total_heads : heads, (half_t *)out);
    } else {
        if (smem > 48 * 1024)
            IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_attention<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        k_attention<false><<<grid, dim3(256), smem, ifa_s(stream)>>>((const half_t *)q, (const uint8_t *)kcache, (const uint8_t *)vcache, n_ctx, q_tokens, prefix_len, heads, kv_heads, head_dim, kq_scale, alibi, alibi_base_head, alibi_total_heads > 0 ? alibi_total_heads : heads, (half_t *)out);
    }
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}
