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
#include <map>
#include <mutex>
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

// rows != nullptr: dynamic batching -- row t is ONE new token of its own query: K/V cache and context length come from
// rows[t] (the query's KV cache set), smem is sized for the longest context of the batch
struct AttnRow { const uint8_t *kc, *vc; int n_ctx, pad; };

template <bool Q8>
__global__ void __launch_bounds__(256) k_attention(const half_t *__restrict__ q, const uint8_t *__restrict__ kc,
                                                   const uint8_t *__restrict__ vc, int n_ctx, int q_tokens,
                                                   int prefix_len, int heads, int kv_heads, int head_dim,
                                                   float kq_scale, int alibi, int alibi_base, int alibi_total,
                                                   half_t *__restrict__ out, const AttnRow *__restrict__ rows = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *S = reinterpret_cast<half_t *>(smem);                       // [n_ctx]
    float *red = reinterpret_cast<float *>(smem + (((size_t)n_ctx * 2 + 15) & ~(size_t)15));  // [8]  (n_ctx: batch maximum)
    const int h = blockIdx.x, t = blockIdx.y;
    if (rows) { kc = rows[t].kc; vc = rows[t].vc; n_ctx = rows[t].n_ctx; prefix_len = n_ctx - 1 - t; }
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


// ---------------------------------------------------------------- prefill (q_tokens >= 4): MFMA tiles
// One workgroup per (32-query tile, head); same three stages and the same rounding points as k_attention
// (S and P are half tensors between the stages, like the reference's Gemm_Alg2 -> SoftMax -> Gemm_Alg2), but
//   S tile  = Q[32 x HD] . K^T          v_mfma_f32_32x32x16_f16, keys of a 32-block = B columns, straight from the cache
//   P       = row softmax of the [32 x n_keys] half tile kept in LDS (a wave per 8 rows)
//   O tile  = P . V                     V blocks transposed through LDS ([dim][key]) so a lane's 8 keys are contiguous
// fp32 accumulation in MFMA order instead of index order (products of halfs are exact either way).
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// 8 consecutive elements e0..e0+7 (e0 % 8 == 0) of kv-row j as halfs
template <bool Q8>
__device__ __forceinline__ half8v kv_load8(const uint8_t *cache, size_t row_bytes, int j, int e0)
{
    if constexpr (Q8) {
        const uint8_t *blk = cache + (size_t)j * row_bytes + (size_t)(e0 >> 5) * 34;
        const uint16_t *p16 = reinterpret_cast<const uint16_t *>(blk);
        const float scale = hbits2f(p16[0]);
        half8v r;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t two = p16[1 + ((e0 & 31) >> 1) + w];
            r[2 * w] = f2h((float)(int)(int8_t)(two & 0xFF) * scale);
            r[2 * w + 1] = f2h((float)(int)(int8_t)(two >> 8) * scale);
        }
        return r;
    } else {
        return __builtin_bit_cast(half8v, *reinterpret_cast<const u32x4v *>(cache + (size_t)j * row_bytes + (size_t)e0 * 2));
    }
}

constexpr int PF_QT = 32;            // queries per workgroup
constexpr int PF_VROW = 40;          // halfs per row of the transposed V block (32 keys + pad: conflict-free b128 reads)

__host__ __device__ inline int pf_nkp(int n_keys) { return (n_keys + 31) / 32 * 32 + 8; }

__device__ unsigned long long g_attn_trace[8];     // wall_clock64 stamps of the longest tile of head 0 (tools/probes)

// SG: the [32 x n_keys] score tile lives in a global workspace (sg_ws: [heads][tiles][32][sg_nkp] halfs) instead of the LDS
// -- contexts past ~2300 keys, where the tile no longer fits the 160 KiB; same stages, same rounding points, the tile is
// written once and walked three times through L2 (a fraction of the K / V traffic of the same tile)
template <int HD, bool Q8, bool SG = false>
__global__ void __launch_bounds__(256) k_attention_mfma(const half_t *__restrict__ q, const uint8_t *__restrict__ kc,
                                                        const uint8_t *__restrict__ vc, int n_ctx, int q_tokens,
                                                        int prefix_len, int heads, int kv_heads, float kq_scale,
                                                        int alibi, int alibi_base, int alibi_total, half_t *__restrict__ out,
                                                        half_t *__restrict__ sg_ws = nullptr, int sg_nkp = 0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 16;          // MFMA k-steps over the head dimension
    constexpr int NT = HD / 32;          // 32-wide output tiles
    // workgroups go to the 8 XCDs round-robin in launch order: give every XCD a fixed eighth of the heads, so a head's
    // K/V rows (re-read by each of its query tiles) stay in ONE 4 MB L2, and start the longest (last) tiles first
    int tile = blockIdx.x, h = blockIdx.y;
    if (heads % 8 == 0) {
        const int w = blockIdx.y * gridDim.x + blockIdx.x, xcd = w & 7, r = w >> 3;
        h = (r / (int)gridDim.x) * 8 + xcd;
        tile = (int)gridDim.x - 1 - r % (int)gridDim.x;
    }
    const int t0 = tile * PF_QT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int kvh = h / (heads / kv_heads);
    const int kv_dim = kv_heads * HD;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const int hoff = kvh * HD;
    const int t_last = min(t0 + PF_QT, q_tokens) - 1;
    const int n_keys = min(n_ctx, prefix_len + t_last + 1);      // causal bound of the whole tile
    const int nkb = (n_keys + 31) / 32;
    const int NKP = SG ? sg_nkp : pf_nkp(n_keys);
    half_t *S = SG ? sg_ws + ((size_t)h * gridDim.x + tile) * PF_QT * (size_t)sg_nkp : reinterpret_cast<half_t *>(smem);     // [32][NKP]
    half_t *Vt = SG ? reinterpret_cast<half_t *>(smem) : S + (size_t)PF_QT * NKP;   // [HD][PF_VROW]
    const float alpha = 1.0f / sqrtf((float)HD) / kq_scale;
    const float mk = alibi ? alibi_slope(h + alibi_base, alibi_total) : 0.0f;

    const bool tracer = h == 0 && tile == (int)gridDim.x - 1 && tid == 0;
    if (tracer) g_attn_trace[0] = wall_clock64();
    // ---- S = half(alpha * Q.K^T) (+ALiBi), key blocks strided over the waves
    {
        half8v qa[KS];
        const int tq = min(t0 + i, q_tokens - 1);
#pragma unroll
        for (int s2 = 0; s2 < KS; s2++)
            qa[s2] = __builtin_bit_cast(half8v, *reinterpret_cast<const u32x4v *>(q + ((size_t)tq * heads + h) * HD + 16 * s2 + 8 * g));
        // two register sets of K fragments: the next block of this wave is in flight while one is multiplied
        half8v ka[KS], kb2[KS];
        auto kload = [&](half8v (&kf)[KS], int kb) {
            const int j = min(32 * kb + i, n_ctx - 1);
#pragma unroll
            for (int s2 = 0; s2 < KS; s2++) kf[s2] = kv_load8<Q8>(kc, row_bytes, j, hoff + 16 * s2 + 8 * g);
        };
        auto kscore = [&](const half8v (&kf)[KS], int kb) {
            f32x16v acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < KS; s2++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s2], kf[s2], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
                half_t sv = f2h(alpha * acc[r]);
                if (alibi) { float a = (float)(32 * kb + i) * mk; sv = f2h(a + h2f(sv)); }
                S[(size_t)row * NKP + 32 * kb + i] = sv;
            }
        };
        if (wave < nkb) kload(ka, wave);
        if (wave + 4 < nkb) kload(kb2, wave + 4);
        for (int kb = wave; kb < nkb; kb += 8) {
            kscore(ka, kb);
            if (kb + 8 < nkb) kload(ka, kb + 8);
            if (kb + 4 < nkb) {
                kscore(kb2, kb + 4);
                if (kb + 12 < nkb) kload(kb2, kb + 12);
            }
        }
    }
    __syncthreads();
    if (tracer) g_attn_trace[1] = wall_clock64();
    // ---- P = softmax rows (Tensor_SoftMax_Alg2_Kernel rounding: half(e), then half(half(e) * 1/sum))
    // 16-byte LDS accesses, 8 scores per lane and step (a 2-byte walk is one LDS round trip per element); the phase is
    // VALU-bound (32 x n_keys exponentials per workgroup), so whole chunks skip the per-element causal mask
    for (int rr = 0; rr < 8; rr++) {
        const int row = 8 * wave + rr;
        const int t = t0 + row;
        half_t *Sr = S + (size_t)row * NKP;
        const int n_valid = t < q_tokens ? min(n_ctx, prefix_len + t + 1) : 0;
        const int nch = 4 * nkb;
        float lmax = -INFINITY;
        for (int c = lane; c < nch; c += 64) {
            const half8v v = *reinterpret_cast<const half8v *>(Sr + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (8 * c + e < n_valid) lmax = fmaxf(lmax, kq_scale * h2f(v[e]));
        }
        lmax = wave_max(lmax);
        float lsum = 0.0f;
        for (int c = lane; c < nch; c += 64) {
            const half8v v = *reinterpret_cast<const half8v *>(Sr + 8 * c);
            half8v o;
            if (8 * c + 8 <= n_valid) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float ev = __expf(kq_scale * h2f(v[e]) - lmax);
                    lsum += ev;
                    o[e] = f2h(ev);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float ev = 0.0f;
                    if (8 * c + e < n_valid) ev = __expf(kq_scale * h2f(v[e]) - lmax);
                    lsum += ev;
                    o[e] = f2h(ev);
                }
            }
            *reinterpret_cast<half8v *>(Sr + 8 * c) = o;
        }
        lsum = wave_sum(lsum);
        const float inv = lsum > 0.0f ? 1.0f / lsum : 0.0f;      // rows past q_tokens: all zero
        const int nch_valid = min(nch, (n_valid + 7) / 8);        // beyond: already zero
        for (int c = lane; c < nch_valid; c += 64) {
            const half8v v = *reinterpret_cast<const half8v *>(Sr + 8 * c);
            half8v o;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = f2h(h2f(v[e]) * inv);
            *reinterpret_cast<half8v *>(Sr + 8 * c) = o;
        }
    }
    if (tracer) g_attn_trace[2] = wall_clock64();
    // ---- O = P.V
    f32x16v oacc;
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[r] = 0.0f;
    constexpr int CH = HD / 8;               // 16-byte chunks per V row
    constexpr int KPI = 256 / CH;            // keys staged per pass of the workgroup
    constexpr int NIT = (32 + KPI - 1) / KPI;
    half8v va[NIT], vb[NIT];                 // V rows of the next two key blocks, requested two blocks ahead
    // lane bits: key parity | key pair (8) | chunk low (4); waves: chunk high, then further key groups.  Lane pairs swap
    // halves of their rows so that every lane stores (even key, odd key) dwords of 4 dims: half the LDS stores of a
    // 2-byte transpose, spread over 32 banks instead of 8
    const int vpar = tid & 1, vpair = (tid >> 1) & 7;
    const int vch = ((tid >> 4) & 3) + 4 * ((tid >> 6) % (CH / 4));
    const int vkey0 = 16 * ((tid >> 6) / (CH / 4)) + 2 * vpair;             // even key of the lane pair within a pass
    auto vload = [&](half8v (&v)[NIT], int kt) {
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int key = min(vkey0 + vpar + it * KPI, 31);
            const int j = min(32 * kt + key, n_ctx - 1);
            v[it] = kv_load8<Q8>(vc, row_bytes, j, hoff + 8 * vch);
        }
    };
    auto vstage = [&](const half8v (&v)[NIT]) {
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const u32x4v w = __builtin_bit_cast(u32x4v, v[it]);
            const uint32_t m0 = vpar ? w[2] : w[0], m1 = vpar ? w[3] : w[1];            // the 4 dims this lane stores
            const uint32_t r0 = __shfl_xor(vpar ? w[0] : w[2], 1), r1 = __shfl_xor(vpar ? w[1] : w[3], 1);
            const uint32_t e0 = vpar ? r0 : m0, e1 = vpar ? r1 : m1;                      // even key's values
            const uint32_t o0 = vpar ? m0 : r0, o1 = vpar ? m1 : r1;                      // odd key's values
            const int key = vkey0 + it * KPI;
            if (key < 32) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(Vt + (size_t)(8 * vch + 4 * vpar) * PF_VROW + key);
                dst[0 * PF_VROW / 2] = (e0 & 0xFFFFu) | (o0 << 16);
                dst[1 * PF_VROW / 2] = (e0 >> 16) | (o0 & 0xFFFF0000u);
                dst[2 * PF_VROW / 2] = (e1 & 0xFFFFu) | (o1 << 16);
                dst[3 * PF_VROW / 2] = (e1 >> 16) | (o1 & 0xFFFF0000u);
            }
        }
    };
    auto pv = [&](int kt) {
        if (wave < NT) {
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const half8v pf = *reinterpret_cast<const half8v *>(S + (size_t)i * NKP + 32 * kt + 16 * s2 + 8 * g);
                const half8v vf = *reinterpret_cast<const half8v *>(Vt + (size_t)(32 * wave + i) * PF_VROW + 16 * s2 + 8 * g);
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf, vf, oacc, 0, 0, 0);
            }
        }
    };
    if (nkb > 0) vload(va, 0);
    if (nkb > 1) vload(vb, 1);
    for (int kt = 0; kt < nkb; kt += 2) {
        __syncthreads();                     // previous block consumed (and, first time, P complete)
        vstage(va);
        if (kt + 2 < nkb) vload(va, kt + 2);
        __syncthreads();
        pv(kt);
        if (kt + 1 < nkb) {
            __syncthreads();
            vstage(vb);
            if (kt + 3 < nkb) vload(vb, kt + 3);
            __syncthreads();
            pv(kt + 1);
        }
    }
    if (tracer) g_attn_trace[3] = wall_clock64();
    if (wave < NT) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (t < q_tokens) out[((size_t)t * heads + h) * HD + 32 * wave + i] = f2h(oacc[r]);
        }
    }
}

// ---------------------------------------------------------------- long prompts (q_tokens >= 128): two passes over the keys
// One workgroup per (128-query tile, head), a wave per 32 queries.  The three-stage kernel above keeps a [32 x n_keys]
// score tile (LDS, past ~2300 keys a global workspace) and re-reads a head's K / V once per 32 queries: at 4096 tokens it
// moves more score-tile and K / V bytes through L2 than the matrix cores can hide (1.2 ms per layer).  Here nothing of size
// n_keys is stored: pass 1 streams the K blocks and keeps each query's running max / sum, pass 2 streams K and V, REcomputes
// the scores, rounds them exactly like the staged kernel (S half, e = half(exp), P = half(e * 1/sum)) and multiplies by V
// from registers.  K / V blocks of 32 keys are staged through LDS once per 128 queries.
//   S^T = K . Q^T: keys are the MFMA rows, so a lane holds 16 keys of ONE query (its column): max / sum / P need no
//                  cross-lane traffic except one exchange between the lane halves at the end of pass 1;
//   O^T = V^T . P^T: P^T is then already the B operand (lane = query column, 8 keys per k-step); a lane's registers hold keys
//                  4g + {0-3, 8-11, 16-19, 24-27}, so the V block is transposed into LDS with its key columns in THAT order
//                  (f2_col) -- the sum over keys does not care.
constexpr int F2_QT = 128;
constexpr int F2_VROW = 40;          // halfs per row of the transposed V block (32 key columns + pad)
__host__ __device__ constexpr int f2_col(int k) { return (k & 16) | (((k >> 2) & 1) << 3) | (k & 3) | (((k >> 3) & 1) << 2); }
template <int HD> __device__ __forceinline__ int f2_swz(int row) { return HD == 128 ? (row & 15) : ((row >> 1) & 7); }

template <int HD, bool Q8>
__global__ void __launch_bounds__(256, 2) k_attention_2pass(const half_t *__restrict__ q, const uint8_t *__restrict__ kc,
                                                         const uint8_t *__restrict__ vc, int n_ctx, int q_tokens,
                                                         int prefix_len, int heads, int kv_heads, float kq_scale,
                                                         int alibi, int alibi_base, int alibi_total, half_t *__restrict__ out)
{
    static_assert(HD == 128 || HD == 64, "two-pass attention: head_dim 64 / 128");
    constexpr int KS = HD / 16, NT = HD / 32, CHK = HD / 8;
    constexpr int NKC = 32 * CHK / 256;              // 16-byte chunks of the K block per thread
    // two buffers each: block kb + 1 is staged while block kb is multiplied (one barrier per block)
    __shared__ __attribute__((aligned(16))) char Ks2[2][32 * HD * 2];             // [key][HD halfs], 16-byte chunks swizzled
    __shared__ __attribute__((aligned(16))) half_t Vt2[2][HD * F2_VROW];         // [dim][key column]
    int tile = blockIdx.x, h = blockIdx.y;
    if (heads % 8 == 0) {            // an XCD keeps a fixed eighth of the heads (their K / V in ONE L2); longest tiles first
        const int w = blockIdx.y * gridDim.x + blockIdx.x, xcd = w & 7, r = w >> 3;
        h = (r / (int)gridDim.x) * 8 + xcd;
        tile = (int)gridDim.x - 1 - r % (int)gridDim.x;
    }
    const int t0 = tile * F2_QT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int kvh = h / (heads / kv_heads);
    const int kv_dim = kv_heads * HD;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const int hoff = kvh * HD;
    const int n_keys = min(n_ctx, prefix_len + min(t0 + F2_QT, q_tokens));        // causal bound of the whole tile
    const int nkb = (n_keys + 31) / 32;
    const int tw = t0 + 32 * wave, tq = tw + i;
    const int n_valid = tq < q_tokens ? min(n_ctx, prefix_len + tq + 1) : 0;      // keys this lane's query may see
    const int wave_keys = tw < q_tokens ? min(n_ctx, prefix_len + min(tw + 32, q_tokens)) : 0;
    const int wave_nkb = (wave_keys + 31) / 32;                                   // blocks past it are fully masked for this wave
    const float alpha = 1.0f / sqrtf((float)HD) / kq_scale;
    const float mk = alibi ? alibi_slope(h + alibi_base, alibi_total) : 0.0f;

    half8v qf[KS];
    {
        const int tqc = min(tq, q_tokens - 1);
#pragma unroll
        for (int s2 = 0; s2 < KS; s2++)
            qf[s2] = __builtin_bit_cast(half8v, *reinterpret_cast<const u32x4v *>(q + ((size_t)tqc * heads + h) * HD + 16 * s2 + 8 * g));
    }
    // ---- staging: K block rows as they are (swizzled chunks), V block transposed with permuted key columns
    half8v kreg[NKC];
    auto kload = [&](int kb) {
#pragma unroll
        for (int it = 0; it < NKC; it++) {
            const int idx = tid + it * 256, row = idx / CHK, c = idx % CHK;
            kreg[it] = kv_load8<Q8>(kc, row_bytes, min(32 * kb + row, n_ctx - 1), hoff + 8 * c);
        }
    };
    auto kstage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NKC; it++) {
            const int idx = tid + it * 256, row = idx / CHK, c = idx % CHK;
            *reinterpret_cast<half8v *>(Ks2[buf] + (size_t)row * (HD * 2) + (size_t)((c ^ f2_swz<HD>(row)) << 4)) = kreg[it];
        }
    };
    constexpr int KPI = 256 / CHK;           // keys staged per pass of the workgroup
    constexpr int NIT = (32 + KPI - 1) / KPI;
    half8v vreg[NIT];
    const int vpar = tid & 1, vpair = (tid >> 1) & 7;
    const int vch = ((tid >> 4) & 3) + 4 * ((tid >> 6) % (CHK / 4));
    const int vkey0 = 16 * ((tid >> 6) / (CHK / 4)) + 2 * vpair;                  // even key of the lane pair within a pass
    auto vload = [&](int kb) {
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int key = min(vkey0 + vpar + it * KPI, 31);
            vreg[it] = kv_load8<Q8>(vc, row_bytes, min(32 * kb + key, n_ctx - 1), hoff + 8 * vch);
        }
    };
    auto swap1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); };   // lane ^ 1 (quad_perm [1, 0, 3, 2])
    auto vstage = [&](int buf) {             // lane pairs trade halves of their rows: (even key, odd key) dwords of 4 dims
        half_t *Vt = Vt2[buf];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const u32x4v w = __builtin_bit_cast(u32x4v, vreg[it]);
            const uint32_t m0 = vpar ? w[2] : w[0], m1 = vpar ? w[3] : w[1];
            const uint32_t r0 = swap1(vpar ? w[0] : w[2]), r1 = swap1(vpar ? w[1] : w[3]);
            const uint32_t e0 = vpar ? r0 : m0, e1 = vpar ? r1 : m1;              // even key's values
            const uint32_t o0 = vpar ? m0 : r0, o1 = vpar ? m1 : r1;              // odd key's values
            const int key = vkey0 + it * KPI;
            if (key < 32) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(Vt + (size_t)(8 * vch + 4 * vpar) * F2_VROW + f2_col(key));
                dst[0 * F2_VROW / 2] = (e0 & 0xFFFFu) | (o0 << 16);
                dst[1 * F2_VROW / 2] = (e0 >> 16) | (o0 & 0xFFFF0000u);
                dst[2 * F2_VROW / 2] = (e1 & 0xFFFFu) | (o1 << 16);
                dst[3 * F2_VROW / 2] = (e1 >> 16) | (o1 & 0xFFFF0000u);
            }
        }
    };
    // scores of key block kb for this wave's 32 queries: x[r] = kq_scale * half(alpha * q.k (+ ALiBi)), key = 32 kb + row(r)
    // scores of key block kb for this wave's 32 queries:  x[r] = half(alpha * q.k (+ ALiBi)) as float,  key = 32 kb + row(r);
    // masked keys: -inf.  exp(kq_scale * x - max) is then exp2(fma(log2(e) * kq_scale, x, c)) with c = -log2(e) * kq_scale * max.
    // (the kernel is VALU-bound -- ~10 instructions per score and pass against 16 MFMAs per 512 scores -- so the mask is
    // applied only in blocks that reach past the wave's first query, and the exponent is one fma + v_exp_f32)
    const float l2e_kq = 1.44269504088896341f * kq_scale;
    const int full_keys = tw < q_tokens ? min(n_ctx, prefix_len + tw + 1) : 0;   // every query of the wave sees keys below this
    auto scores = [&](int kb, float (&x)[16]) {
        const char *Ks = Ks2[kb & 1];
        f32x16v acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < KS; s2++) {
            const half8v kf = *reinterpret_cast<const half8v *>(Ks + (size_t)i * (HD * 2) + (size_t)(((2 * s2 + g) ^ f2_swz<HD>(i)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s2], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            half_t sv = f2h(alpha * acc[r]);
            if (alibi) sv = f2h((float)(32 * kb + (r & 3) + 8 * (r >> 2) + 4 * g) * mk + h2f(sv));
            x[r] = h2f(sv);
        }
        if (32 * kb + 32 > full_keys) {                   // (wave-uniform: only the blocks on the causal edge)
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (32 * kb + (r & 3) + 8 * (r >> 2) + 4 * g >= n_valid) x[r] = -INFINITY;
        }
    };
    auto exp2_fast = [](float v) { return __builtin_amdgcn_exp2f(v); };

    // ---- pass 1: running max / sum of every query over its keys
    float m = -INFINITY, sum = 0.0f;
    kload(0);
    kstage(0);
    kload(min(1, nkb - 1));
    __syncthreads();
    for (int kb = 0; kb < nkb; kb++) {
        kstage((kb + 1) & 1);                // (block kb + 1, or a harmless repeat of the last one)
        kload(min(kb + 2, nkb - 1));
        if (kb < wave_nkb) {
            float x[16];
            scores(kb, x);
            float bm = x[0];
#pragma unroll
            for (int r = 1; r < 16; r++) bm = fmaxf(bm, x[r]);
            if (bm > m) { sum *= exp2_fast(l2e_kq * (m - bm)); m = bm; }             // (m = -inf: sum is 0 and stays 0)
            if (m > -INFINITY) {
                const float c = -l2e_kq * m;
#pragma unroll
                for (int r = 0; r < 16; r++) sum += exp2_fast(__builtin_fmaf(l2e_kq, x[r], c));   // masked: 2^-inf = 0
            }
        }
        __syncthreads();
    }
    float M, inv;
    {
        const float m2 = __shfl_xor(m, 32), s2v = __shfl_xor(sum, 32);
        M = fmaxf(m, m2);
        const float tot = (m > -INFINITY ? sum * exp2_fast(l2e_kq * (m - M)) : 0.0f) + (m2 > -INFINITY ? s2v * exp2_fast(l2e_kq * (m2 - M)) : 0.0f);
        inv = tot > 0.0f ? 1.0f / tot : 0.0f;
    }
    // ---- pass 2: P = half(half(exp(x - M)) / sum) from recomputed scores, O^T += V^T . P^T
    f32x16v oacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[nt][r] = 0.0f;
    kload(0);
    vload(0);
    kstage(0);
    vstage(0);
    kload(min(1, nkb - 1));
    vload(min(1, nkb - 1));
    __syncthreads();
    for (int kb = 0; kb < nkb; kb++) {
        kstage((kb + 1) & 1);
        vstage((kb + 1) & 1);
        kload(min(kb + 2, nkb - 1));
        vload(min(kb + 2, nkb - 1));
        if (kb < wave_nkb) {
            const half_t *Vt = Vt2[kb & 1];
            float x[16];
            scores(kb, x);
            half8v pb[2];
            const float c = M > -INFINITY ? -l2e_kq * M : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const half_t e = f2h(exp2_fast(__builtin_fmaf(l2e_kq, x[r], c)));       // masked keys: 2^-inf = 0
                pb[r >> 3][r & 7] = f2h(h2f(e) * inv);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const half8v vf = *reinterpret_cast<const half8v *>(Vt + (size_t)(32 * nt + i) * F2_VROW + 16 * s2 + 8 * g);
                    oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pb[s2], oacc[nt], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    if (tq < q_tokens) {
        half_t *orow = out + ((size_t)tq * heads + h) * HD;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                typedef _Float16 half4v __attribute__((ext_vector_type(4)));
                half4v o4;
#pragma unroll
                for (int e = 0; e < 4; e++) o4[e] = f2h(oacc[nt][4 * j + e]);
                *reinterpret_cast<half4v *>(orow + 32 * nt + 8 * j + 4 * g) = o4;
            }
    }
}

template <int HD, bool Q8>
__global__ void __launch_bounds__(256, 2) k_attention_2pass_ks(const half_t *__restrict__ q, const uint8_t *__restrict__ kc,
                                                         const uint8_t *__restrict__ vc, int n_ctx, int q_tokens,
                                                         int prefix_len, int heads, int kv_heads, float kq_scale,
                                                         int alibi, int alibi_base, int alibi_total, half_t *__restrict__ out)
{
    static_assert(HD == 128 || HD == 64, "two-pass attention: head_dim 64 / 128");
    constexpr int KS = HD / 16, NT = HD / 32, CHK = HD / 8;
    constexpr int NKC = 32 * CHK / 256;              // 16-byte chunks of the K block per thread
    // 64 queries per workgroup: waves 0 / 1 take the even key blocks for queries 0-31 / 32-63, waves 2 / 3 the odd ones; a
    // stage is TWO key blocks (one barrier per 64 keys), the halves' running max / sum and output tiles meet through LDS
    constexpr int KBYTES = 32 * HD * 2, VHALFS = HD * F2_VROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];                   // [2 stages][2 parities] K blocks, then V blocks
    auto KsP = [&](int buf, int p) { return smem + (size_t)(buf * 2 + p) * KBYTES; };
    auto VtP = [&](int buf, int p) { return reinterpret_cast<half_t *>(smem + (size_t)4 * KBYTES) + (size_t)(buf * 2 + p) * VHALFS; };
    int tile = blockIdx.x, h = blockIdx.y;
    if (heads % 8 == 0) {            // an XCD keeps a fixed eighth of the heads (their K / V in ONE L2); longest tiles first
        const int w = blockIdx.y * gridDim.x + blockIdx.x, xcd = w & 7, r = w >> 3;
        h = (r / (int)gridDim.x) * 8 + xcd;
        tile = (int)gridDim.x - 1 - r % (int)gridDim.x;
    }
    const int t0 = tile * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = wave & 1, kp = wave >> 1;          // query half, key-block parity
    const int i = lane & 31, g = lane >> 5;
    const int kvh = h / (heads / kv_heads);
    const int kv_dim = kv_heads * HD;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const int hoff = kvh * HD;
    const int n_keys = min(n_ctx, prefix_len + min(t0 + 64, q_tokens));           // causal bound of the whole tile
    const int nkb = (n_keys + 31) / 32, nit = (nkb + 1) / 2;
    const int tw = t0 + 32 * qw, tq = tw + i;
    const int n_valid = tq < q_tokens ? min(n_ctx, prefix_len + tq + 1) : 0;      // keys this lane's query may see
    const int wave_keys = tw < q_tokens ? min(n_ctx, prefix_len + min(tw + 32, q_tokens)) : 0;
    const int wave_nkb = (wave_keys + 31) / 32;                                   // blocks past it are fully masked for this wave
    const float alpha = 1.0f / sqrtf((float)HD) / kq_scale;
    const float mk = alibi ? alibi_slope(h + alibi_base, alibi_total) : 0.0f;

    half8v qf[KS];
    {
        const int tqc = min(tq, q_tokens - 1);
#pragma unroll
        for (int s2 = 0; s2 < KS; s2++)
            qf[s2] = __builtin_bit_cast(half8v, *reinterpret_cast<const u32x4v *>(q + ((size_t)tqc * heads + h) * HD + 16 * s2 + 8 * g));
    }
    // ---- staging: K block rows as they are (swizzled chunks), V block transposed with permuted key columns
    half8v kreg[2][NKC];
    auto kload = [&](int itn) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int kb = min(2 * itn + p, nkb - 1);
#pragma unroll
            for (int it = 0; it < NKC; it++) {
                const int idx = tid + it * 256, row = idx / CHK, c = idx % CHK;
                kreg[p][it] = kv_load8<Q8>(kc, row_bytes, min(32 * kb + row, n_ctx - 1), hoff + 8 * c);
            }
        }
    };
    auto kstage = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int it = 0; it < NKC; it++) {
                const int idx = tid + it * 256, row = idx / CHK, c = idx % CHK;
                *reinterpret_cast<half8v *>(KsP(buf, p) + (size_t)row * (HD * 2) + (size_t)((c ^ f2_swz<HD>(row)) << 4)) = kreg[p][it];
            }
    };
    constexpr int KPI = 256 / CHK;           // keys staged per pass of the workgroup
    constexpr int NIT = (32 + KPI - 1) / KPI;
    half8v vreg[2][NIT];
    const int vpar = tid & 1, vpair = (tid >> 1) & 7;
    const int vch = ((tid >> 4) & 3) + 4 * ((tid >> 6) % (CHK / 4));
    const int vkey0 = 16 * ((tid >> 6) / (CHK / 4)) + 2 * vpair;                  // even key of the lane pair within a pass
    auto vload = [&](int itn) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int kb = min(2 * itn + p, nkb - 1);
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int key = min(vkey0 + vpar + it * KPI, 31);
                vreg[p][it] = kv_load8<Q8>(vc, row_bytes, min(32 * kb + key, n_ctx - 1), hoff + 8 * vch);
            }
        }
    };
    auto swap1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); };   // lane ^ 1 (quad_perm [1, 0, 3, 2])
    auto vstage = [&](int buf) {             // lane pairs trade halves of their rows: (even key, odd key) dwords of 4 dims
#pragma unroll
      for (int p = 0; p < 2; p++) {
        half_t *Vt = VtP(buf, p);
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const u32x4v w = __builtin_bit_cast(u32x4v, vreg[p][it]);
            const uint32_t m0 = vpar ? w[2] : w[0], m1 = vpar ? w[3] : w[1];
            const uint32_t r0 = swap1(vpar ? w[0] : w[2]), r1 = swap1(vpar ? w[1] : w[3]);
            const uint32_t e0 = vpar ? r0 : m0, e1 = vpar ? r1 : m1;              // even key's values
            const uint32_t o0 = vpar ? m0 : r0, o1 = vpar ? m1 : r1;              // odd key's values
            const int key = vkey0 + it * KPI;
            if (key < 32) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(Vt + (size_t)(8 * vch + 4 * vpar) * F2_VROW + f2_col(key));
                dst[0 * F2_VROW / 2] = (e0 & 0xFFFFu) | (o0 << 16);
                dst[1 * F2_VROW / 2] = (e0 >> 16) | (o0 & 0xFFFF0000u);
                dst[2 * F2_VROW / 2] = (e1 & 0xFFFFu) | (o1 << 16);
                dst[3 * F2_VROW / 2] = (e1 >> 16) | (o1 & 0xFFFF0000u);
            }
        }
      }
    };
    // scores of key block kb for this wave's 32 queries: x[r] = kq_scale * half(alpha * q.k (+ ALiBi)), key = 32 kb + row(r)
    // scores of key block kb for this wave's 32 queries:  x[r] = half(alpha * q.k (+ ALiBi)) as float,  key = 32 kb + row(r);
    // masked keys: -inf.  exp(kq_scale * x - max) is then exp2(fma(log2(e) * kq_scale, x, c)) with c = -log2(e) * kq_scale * max.
    // (the kernel is VALU-bound -- ~10 instructions per score and pass against 16 MFMAs per 512 scores -- so the mask is
    // applied only in blocks that reach past the wave's first query, and the exponent is one fma + v_exp_f32)
    const float l2e_kq = 1.44269504088896341f * kq_scale;
    const int full_keys = tw < q_tokens ? min(n_ctx, prefix_len + tw + 1) : 0;   // every query of the wave sees keys below this
    auto scores = [&](int kb, int buf, float (&x)[16]) {
        const char *Ks = KsP(buf, kp);
        f32x16v acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < KS; s2++) {
            const half8v kf = *reinterpret_cast<const half8v *>(Ks + (size_t)i * (HD * 2) + (size_t)(((2 * s2 + g) ^ f2_swz<HD>(i)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s2], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            half_t sv = f2h(alpha * acc[r]);
            if (alibi) sv = f2h((float)(32 * kb + (r & 3) + 8 * (r >> 2) + 4 * g) * mk + h2f(sv));
            x[r] = h2f(sv);
        }
        if (32 * kb + 32 > full_keys) {                   // (wave-uniform: only the blocks on the causal edge)
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (32 * kb + (r & 3) + 8 * (r >> 2) + 4 * g >= n_valid) x[r] = -INFINITY;
        }
    };
    auto exp2_fast = [](float v) { return __builtin_amdgcn_exp2f(v); };

    // ---- pass 1: running max / sum of every query over its keys
    float m = -INFINITY, sum = 0.0f;
    kload(0);
    kstage(0);
    kload(min(1, nit - 1));
    __syncthreads();
    for (int itn = 0; itn < nit; itn++) {
        const int kb = 2 * itn + kp;
        kstage((itn + 1) & 1);               // (the next pair of blocks, or a harmless repeat of the last one)
        kload(min(itn + 2, nit - 1));
        if (kb < wave_nkb && kb < nkb) {
            float x[16];
            scores(kb, itn & 1, x);
            float bm = x[0];
#pragma unroll
            for (int r = 1; r < 16; r++) bm = fmaxf(bm, x[r]);
            if (bm > m) { sum *= exp2_fast(l2e_kq * (m - bm)); m = bm; }             // (m = -inf: sum is 0 and stays 0)
            if (m > -INFINITY) {
                const float c = -l2e_kq * m;
#pragma unroll
                for (int r = 0; r < 16; r++) sum += exp2_fast(__builtin_fmaf(l2e_kq, x[r], c));   // masked: 2^-inf = 0
            }
        }
        __syncthreads();
    }
    float M, inv;
    {
        const float m2 = __shfl_xor(m, 32), s2v = __shfl_xor(sum, 32);
        float Mw = fmaxf(m, m2);
        float totw = (m > -INFINITY ? sum * exp2_fast(l2e_kq * (m - Mw)) : 0.0f) + (m2 > -INFINITY ? s2v * exp2_fast(l2e_kq * (m2 - Mw)) : 0.0f);
        // the two key parities of a query meet through LDS (the staging area is idle between the passes)
        float *ms = reinterpret_cast<float *>(smem);             // [query half][32][2]
        if (kp == 1 && g == 0) { ms[(qw * 32 + i) * 2] = Mw; ms[(qw * 32 + i) * 2 + 1] = totw; }
        __syncthreads();
        M = Mw; inv = 0.0f;
        if (kp == 0) {
            const float m1 = ms[(qw * 32 + i) * 2], t1 = ms[(qw * 32 + i) * 2 + 1];
            M = fmaxf(Mw, m1);
            const float tot = (Mw > -INFINITY ? totw * exp2_fast(l2e_kq * (Mw - M)) : 0.0f) + (m1 > -INFINITY ? t1 * exp2_fast(l2e_kq * (m1 - M)) : 0.0f);
            inv = tot > 0.0f ? 1.0f / tot : 0.0f;
        }
        __syncthreads();
        if (kp == 0 && g == 0) { ms[(qw * 32 + i) * 2] = M; ms[(qw * 32 + i) * 2 + 1] = inv; }
        __syncthreads();
        if (kp == 1) { M = ms[(qw * 32 + i) * 2]; inv = ms[(qw * 32 + i) * 2 + 1]; }
        __syncthreads();
    }
    // ---- pass 2: P = half(half(exp(x - M)) / sum) from recomputed scores, O^T += V^T . P^T
    f32x16v oacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[nt][r] = 0.0f;
    kload(0);
    vload(0);
    kstage(0);
    vstage(0);
    kload(min(1, nit - 1));
    vload(min(1, nit - 1));
    __syncthreads();
    for (int itn = 0; itn < nit; itn++) {
        const int kb = 2 * itn + kp;
        kstage((itn + 1) & 1);
        vstage((itn + 1) & 1);
        kload(min(itn + 2, nit - 1));
        vload(min(itn + 2, nit - 1));
        if (kb < wave_nkb && kb < nkb) {
            const half_t *Vt = VtP(itn & 1, kp);
            float x[16];
            scores(kb, itn & 1, x);
            half8v pb[2];
            const float c = M > -INFINITY ? -l2e_kq * M : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const half_t e = f2h(exp2_fast(__builtin_fmaf(l2e_kq, x[r], c)));       // masked keys: 2^-inf = 0
                pb[r >> 3][r & 7] = f2h(h2f(e) * inv);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const half8v vf = *reinterpret_cast<const half8v *>(Vt + (size_t)(32 * nt + i) * F2_VROW + 16 * s2 + 8 * g);
                    oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pb[s2], oacc[nt], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    {   // the odd-parity half hands its output tiles over (staging area: idle after the last barrier)
        float *os = reinterpret_cast<float *>(smem);             // [query half][NT][16][64]
        if (kp == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) os[((qw * NT + nt) * 16 + r) * 64 + lane] = oacc[nt][r];
        }
        __syncthreads();
        if (kp == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) oacc[nt][r] += os[((qw * NT + nt) * 16 + r) * 64 + lane];
        }
    }
    if (kp == 0 && tq < q_tokens) {
        half_t *orow = out + ((size_t)tq * heads + h) * HD;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                typedef _Float16 half4v __attribute__((ext_vector_type(4)));
                half4v o4;
#pragma unroll
                for (int e = 0; e < 4; e++) o4[e] = f2h(oacc[nt][4 * j + e]);
                *reinterpret_cast<half4v *>(orow + 32 * nt + 8 * j + 4 * g) = o4;
            }
    }
}

} // namespace ifa

using namespace ifa;

template <int HD>
static int launch_attention_mfma(const void *q, const void *kcache, const void *vcache, int kv_dtype, int n_ctx, int q_tokens,
                                 int prefix_len, int heads, int kv_heads, float kq_scale, int alibi, int alibi_base,
                                 int alibi_total, void *out, size_t smem, hipStream_t s, half_t *sg_ws = nullptr, int sg_nkp = 0)
{
    dim3 grid((unsigned)((q_tokens + PF_QT - 1) / PF_QT), (unsigned)heads);
#define IFA_PFA(Q8V, SGV) { auto kern = k_attention_mfma<HD, Q8V, SGV>; \
        if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        kern<<<grid, dim3(256), smem, s>>>((const half_t *)q, (const uint8_t *)kcache, (const uint8_t *)vcache, n_ctx, q_tokens, \
                                           prefix_len, heads, kv_heads, kq_scale, alibi, alibi_base, alibi_total, (half_t *)out, sg_ws, sg_nkp); }
    if (kv_dtype == Q8_B32T2) { if (sg_ws) IFA_PFA(true, true) else IFA_PFA(true, false) }
    else { if (sg_ws) IFA_PFA(false, true) else IFA_PFA(false, false) }
#undef IFA_PFA
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// workspace of the global-score-tile variant, one per (device, stream), grown on demand (growth synchronises the stream)
namespace {
struct SgWs { void *p = nullptr; size_t bytes = 0; };
std::mutex g_sg_mutex;
std::map<std::pair<int, hipStream_t>, SgWs> g_sg_ws;
}
namespace ifa {
void attn_release_stream(int dev, hipStream_t s)          // called by ifa_gemm_release_stream (ifa_gemm.hip)
{
    std::lock_guard<std::mutex> lock(g_sg_mutex);
    auto it = g_sg_ws.find({dev, s});
    if (it == g_sg_ws.end()) return;
    if (it->second.p) (void)hipFree(it->second.p);
    g_sg_ws.erase(it);
}
}
static int sg_workspace(hipStream_t s, size_t bytes, half_t **out)
{
    int dev = 0;
    IFA_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_sg_mutex);
    SgWs &w = g_sg_ws[{dev, s}];
    if (bytes > w.bytes) {
        if (w.p) { IFA_HIP_CHECK(hipStreamSynchronize(s)); IFA_HIP_CHECK(hipFree(w.p)); w.p = nullptr; w.bytes = 0; }
        IFA_HIP_CHECK(hipMalloc(&w.p, bytes));
        w.bytes = bytes;
    }
    *out = (half_t *)w.p;
    return IFA_OK;
}

static int g_attn_2pass_min = 64;        // chunks of at least this many queries take the two-pass kernels (ifa_attention_two_pass_min)
static int g_attn_2pass_min_keys = 1 << 30;      // contexts of at least this many keys: the 128-query variant instead of the 64-query one (never by default;
                                                 // 128 / 256 / 512 / 1024 / 2048 / 4096 tokens: 64-query 11.9 / 16.9 / 27.4 / 60.2 / 165 / 492 us, staged kernel 15.6 / 18.9 / 32.7 / 77 / 334 / 1202,
                                                 // 128-query variant - / - / - / 75 / 175 / 519 -- Llama-2-7B heads, profiles/r02_attention_two_pass.log)

extern "C" int ifa_attention_two_pass_min(int min_tokens)
{
    const int prev = g_attn_2pass_min;
    if (min_tokens >= 0) g_attn_2pass_min = min_tokens > 0 ? min_tokens : (1 << 30);
    return prev;
}

extern "C" int ifa_attention_two_pass_min_keys(int min_keys)
{
    const int prev = g_attn_2pass_min_keys;
    if (min_keys >= 0) g_attn_2pass_min_keys = min_keys;
    return prev;
}

extern "C" int ifa_debug_attn_trace(unsigned long long *out8)
{
    IFA_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(ifa::g_attn_trace), 64));
    return IFA_OK;
}

// engine-internal: n query rows, each one token on its own KV cache (rows_dev[t]); max_ctx = longest context
extern "C" int ifa_attention_rows(const void *q, const void *rows_dev, int kv_dtype, int n_rows, int max_ctx, int heads, int kv_heads,
                                  int head_dim, float kq_scale, int alibi, int alibi_base_head, int alibi_total_heads, void *out,
                                  ifa_stream stream)
{
    IFA_REQUIRE(q && rows_dev && out && n_rows > 0 && max_ctx > 0, "ifa_attention_rows: bad arguments");
    const size_t smem = (((size_t)max_ctx * 2 + 15) & ~(size_t)15) + 64;
    dim3 grid((unsigned)heads, (unsigned)n_rows);
    const int total_heads = alibi_total_heads > 0 ? alibi_total_heads : heads;
    if (kv_dtype == Q8_B32T2) {
        if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_attention<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        k_attention<true><<<grid, dim3(256), smem, ifa_s(stream)>>>((const half_t *)q, nullptr, nullptr, max_ctx, n_rows, 0, heads, kv_heads, head_dim, kq_scale,
                                                                    alibi, alibi_base_head, total_heads, (half_t *)out, (const AttnRow *)rows_dev);
    } else {
        if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_attention<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        k_attention<false><<<grid, dim3(256), smem, ifa_s(stream)>>>((const half_t *)q, nullptr, nullptr, max_ctx, n_rows, 0, heads, kv_heads, head_dim, kq_scale,
                                                                     alibi, alibi_base_head, total_heads, (half_t *)out, (const AttnRow *)rows_dev);
    }
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

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
    const int total_heads = alibi_total_heads > 0 ? alibi_total_heads : heads;
    // chunks of >= 64 queries: the two-pass kernel with 64 queries per workgroup and the key blocks split over wave pairs
    if (q_tokens >= g_attn_2pass_min && !(q_tokens >= 128 && n_ctx >= g_attn_2pass_min_keys) && (head_dim == 64 || head_dim == 128)) {
        hipStream_t hs = ifa_s(stream);
        dim3 grid((unsigned)((q_tokens + 63) / 64), (unsigned)heads);
        const size_t smem = (size_t)4 * 32 * head_dim * 2 + (size_t)4 * head_dim * F2_VROW * 2;
#define IFA_F2K(HDV, Q8V) { auto kern = k_attention_2pass_ks<HDV, Q8V>; \
        if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        kern<<<grid, dim3(256), smem, hs>>>((const half_t *)q, (const uint8_t *)kcache, (const uint8_t *)vcache, n_ctx, q_tokens, \
                                            prefix_len, heads, kv_heads, kq_scale, alibi, alibi_base_head, total_heads, (half_t *)out); }
        if (head_dim == 128) { if (kv_dtype == Q8_B32T2) IFA_F2K(128, true) else IFA_F2K(128, false) }
        else { if (kv_dtype == Q8_B32T2) IFA_F2K(64, true) else IFA_F2K(64, false) }
#undef IFA_F2K
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    // the 128-query variant of it (one wave per 32 queries, every wave walks all key blocks): behind the 64-query one at every
    // length measured (4096 tokens: 519 vs 492 us), selectable through ifa_attention_two_pass_min_keys
    if (q_tokens >= 128 && q_tokens >= g_attn_2pass_min && n_ctx >= g_attn_2pass_min_keys && (head_dim == 64 || head_dim == 128)) {
        hipStream_t hs = ifa_s(stream);
        dim3 grid((unsigned)((q_tokens + F2_QT - 1) / F2_QT), (unsigned)heads);
#define IFA_F2A(HDV, Q8V) k_attention_2pass<HDV, Q8V><<<grid, dim3(256), 0, hs>>>((const half_t *)q, (const uint8_t *)kcache, (const uint8_t *)vcache, n_ctx, q_tokens, \
                                                                                 prefix_len, heads, kv_heads, kq_scale, alibi, alibi_base_head, total_heads, (half_t *)out)
        if (head_dim == 128) { if (kv_dtype == Q8_B32T2) IFA_F2A(128, true); else IFA_F2A(128, false); }
        else { if (kv_dtype == Q8_B32T2) IFA_F2A(64, true); else IFA_F2A(64, false); }
#undef IFA_F2A
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    // prefill: MFMA tiles whenever the [32 x n_keys] score tile fits the 160 KiB LDS (~2300 keys)
    const size_t smem_mfma = (size_t)PF_QT * pf_nkp(n_ctx) * 2 + (size_t)head_dim * PF_VROW * 2 + 64;
    if (q_tokens >= 4 && smem_mfma <= 150 * 1024 && (head_dim == 32 || head_dim == 64 || head_dim == 128)) {
        hipStream_t hs = ifa_s(stream);
        switch (head_dim) {
        case 32: return launch_attention_mfma<32>(q, kcache, vcache, kv_dtype, n_ctx, q_tokens, prefix_len, heads, kv_heads, kq_scale, alibi, alibi_base_head, total_heads, out, smem_mfma, hs);
        case 64: return launch_attention_mfma<64>(q, kcache, vcache, kv_dtype, n_ctx, q_tokens, prefix_len, heads, kv_heads, kq_scale, alibi, alibi_base_head, total_heads, out, smem_mfma, hs);
        default: return launch_attention_mfma<128>(q, kcache, vcache, kv_dtype, n_ctx, q_tokens, prefix_len, heads, kv_heads, kq_scale, alibi, alibi_base_head, total_heads, out, smem_mfma, hs);
        }
    }
    if (q_tokens >= 4 && (head_dim == 32 || head_dim == 64 || head_dim == 128)) {
        // longer contexts: the same kernel with its score tiles in a global workspace (the op-level kernel below takes 131 ms
        // per layer for 2048 queries on 4096 keys)
        const int nkp = pf_nkp(n_ctx);
        const size_t tiles = (size_t)((q_tokens + PF_QT - 1) / PF_QT);
        half_t *ws = nullptr;
        int rc = sg_workspace(ifa_s(stream), (size_t)heads * tiles * PF_QT * (size_t)nkp * 2, &ws);
        if (rc) return rc;
        const size_t smem_v = (size_t)head_dim * PF_VROW * 2 + 64;
        hipStream_t hs = ifa_s(stream);
        switch (head_dim) {
        case 32: return launch_attention_mfma<32>(q, kcache, vcache, kv_dtype, n_ctx, q_tokens, prefix_len, heads, kv_heads, kq_scale, alibi, alibi_base_head, total_heads, out, smem_v, hs, ws, nkp);
        case 64: return launch_attention_mfma<64>(q, kcache, vcache, kv_dtype, n_ctx, q_tokens, prefix_len, heads, kv_heads, kq_scale, alibi, alibi_base_head, total_heads, out, smem_v, hs, ws, nkp);
        default: return launch_attention_mfma<128>(q, kcache, vcache, kv_dtype, n_ctx, q_tokens, prefix_len, heads, kv_heads, kq_scale, alibi, alibi_base_head, total_heads, out, smem_v, hs, ws, nkp);
        }
    }
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
