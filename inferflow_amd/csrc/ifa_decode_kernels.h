// ifa_decode_kernels.h -- fused batch-1 decode kernels (the hot path).
//
// One decoder layer at T=1 is 5 launches instead of the reference's ~30
// launch+sync pairs:
//   k_dec_gemv<QKV>   : [RMSNorm -> Q8 act-quant] prologue + Wq/Wk/Wv GEMV (+bias)
//   k_dec_attn        : RoPE(q,k) + KV-cache write (F16 or Q8) + GQA attention
//   k_dec_gemv<WO>    : [Q8 act-quant] + Wo GEMV + bias + residual add
//   k_dec_gemv<FFN13> : [RMSNorm -> quant] + W1,W3 GEMV + act(t1)*t2
//   k_dec_gemv<W2>    : [quant] + W2 GEMV + bias + residual add(s)
// plus embedding gather, final-norm + F16 lm_head GEMV and a device-side greedy
// argmax that also advances the position, so a whole step is graph-replayable.
//
// Rounding points are the reference's (every op boundary is an F16 tensor); the
// prologue reproduces Tensor_RmsNorm_Kernel's partial-sum order exactly, so the
// int8 activation codes are bit-identical to the op-by-op path.
//
// Weight rows are streamed from the row-local plane layout (ifa_tiled.h):
// lane l owns blocks l+64j of every row, issues one aligned 16-byte load per
// block plus a 4-byte (base,scale) load, and keeps its slice of the int8
// activation in registers for all rows it processes.  The first batch of weight
// loads is issued BEFORE the prologue so HBM latency overlaps the norm/quant.
#pragma once
#include "ifa_device.h"
#include "ifa_math.h"

namespace ifa {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

constexpr int DEC_THREADS = 512;   // 8 waves per workgroup

// ------------------------------------------------------------------ LDS image
// of the (normalised and) quantised activation vector
struct XLds {
    int8_t *codes;   // [cols], natural element order, 16-byte aligned
    float *scale;    // [cols/32]  fp32 value of the fp16-rounded block scale
    float *xsum;     // [cols/32]  sum of the 32 int8 codes (exact in fp32)
    float *part;     // [128] partial sums + [4] stats
};

__host__ __device__ inline size_t xlds_bytes(int cols)
{
    size_t nb = (size_t)cols / 32;
    return ((size_t)cols + 15) / 16 * 16 + nb * 8 + 132 * 4 + 16;
}

__device__ __forceinline__ XLds xlds_carve(char *smem, int cols)
{
    XLds l;
    size_t nb = (size_t)cols / 32;
    size_t off = 0;
    l.codes = reinterpret_cast<int8_t *>(smem); off += ((size_t)cols + 15) / 16 * 16;
    l.scale = reinterpret_cast<float *>(smem + off); off += nb * 4;
    l.xsum = reinterpret_cast<float *>(smem + off); off += nb * 4;
    off = (off + 15) / 16 * 16;
    l.part = reinterpret_cast<float *>(smem + off);
    return l;
}

// Prologue shared by every decode GEMV kernel.  NORM: 0 none, 1 RMS.
//   xn = NORM ? half(rms(x)) : x ;  Q8_B32T2 quantisation of xn exactly as
//   Tensor_QuantizeQ8_B32T2_Alg2_Kernel (src/kernels/tensor_quant.h:44-82).
// Must be called by all DEC_THREADS threads.  cols % 32 == 0.
template <int NORM>
__device__ __forceinline__ void dec_prologue(const half_t *__restrict__ x, const half_t *__restrict__ nw,
                                             const half_t *__restrict__ nb, float multi_base, float eps, int cols,
                                             const XLds &L, half_t *__restrict__ xn_out)
{
    const int tid = threadIdx.x;
    float scale = 1.0f;
    if constexpr (NORM == 1) {
        if (tid < 128) L.part[tid] = rms_partial(x, cols, tid, 128);
        __syncthreads();
        if (tid == 0) L.part[128] = rms_scale_from_partials(L.part, 128, cols, eps);
        __syncthreads();
        scale = L.part[128];
    }
    const int chunks = cols >> 3;
    for (int c = tid; c < chunks; c += DEC_THREADS) {
        const half8_t xv = *reinterpret_cast<const half8_t *>(x + (size_t)c * 8);
        float v[8];
        if constexpr (NORM == 1) {
            half8_t wv, bv;
            if (nw) wv = *reinterpret_cast<const half8_t *>(nw + (size_t)c * 8);
            if (nb) bv = *reinterpret_cast<const half8_t *>(nb + (size_t)c * 8);
            half8_t outv;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float t = (float)xv[i] * scale;
                if (nw) {
                    float m = multi_base + (float)wv[i];
                    t = t * m;
                    if (nb) t = t + (float)bv[i];
                }
                half_t th = f2h(t);
                outv[i] = th;
                v[i] = h2f(th);
            }
            if (xn_out && blockIdx.x == 0 && blockIdx.y == 0) *reinterpret_cast<half8_t *>(xn_out + (size_t)c * 8) = outv;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (float)xv[i];
        }
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
        mx = fmaxf(mx, dpp_xor1(mx));     // 4 consecutive lanes == one 32-element block
        mx = fmaxf(mx, dpp_xor2(mx));
        const float qs = mx / 127;
        int q[8]; int s = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int qq = qs <= 0.000001f ? 0 : (int)roundf(v[i] / qs);
            qq = min(max(qq, -128), 127);
            q[i] = qq; s += qq;
        }
        s += dpp_xor1(s);
        s += dpp_xor2(s);
        u32x2 packed;
        packed[0] = (uint32_t)(q[0] & 0xFF) | ((uint32_t)(q[1] & 0xFF) << 8) | ((uint32_t)(q[2] & 0xFF) << 16) | ((uint32_t)(q[3] & 0xFF) << 24);
        packed[1] = (uint32_t)(q[4] & 0xFF) | ((uint32_t)(q[5] & 0xFF) << 8) | ((uint32_t)(q[6] & 0xFF) << 16) | ((uint32_t)(q[7] & 0xFF) << 24);
        *reinterpret_cast<u32x2 *>(L.codes + (size_t)c * 8) = packed;
        if ((c & 3) == 0) {
            L.scale[c >> 2] = h2f(f2h(qs));    // the fp16-rounded scale is what the GEMV multiplies by
            L.xsum[c >> 2] = (float)s;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ Q4_B32T1
// registers of one lane for NJ blocks of the activation / of one weight row
template <int NJ>
struct XRegsQ4 {
    int xe[NJ][4], xo[NJ][4];
    float xs[NJ], xsf[NJ];
    __device__ __forceinline__ void load(const XLds &L, int lane, int nblk)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = lane + 64 * j;
            xs[j] = 0.0f; xsf[j] = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; w++) { xe[j][w] = 0; xo[j][w] = 0; }
            if (blk < nblk) {
                const u32x4 a = *reinterpret_cast<const u32x4 *>(L.codes + (size_t)blk * 32);
                const u32x4 b = *reinterpret_cast<const u32x4 *>(L.codes + (size_t)blk * 32 + 16);
                const uint32_t d[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    xe[j][w] = (int)__builtin_amdgcn_perm(d[2 * w + 1], d[2 * w], 0x06040200u);
                    xo[j][w] = (int)__builtin_amdgcn_perm(d[2 * w + 1], d[2 * w], 0x07050301u);
                }
                xs[j] = L.scale[blk];
                xsf[j] = L.xsum[blk];
            }
        }
    }
};

template <int NJ>
struct WRowQ4 {
    u32x4 c[NJ];
    uint32_t sb[NJ];
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane, bool row_ok)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = lane + 64 * j;
            c[j] = u32x4{0, 0, 0, 0}; sb[j] = 0;
            if (row_ok && blk < nblk) {
                c[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow + (size_t)blk * 16));
                sb[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(wrow + (size_t)nblk * 16 + (size_t)blk * 4));
            }
        }
    }
    // lane-partial of sum_blk xs*(dot*scale + xsum*base); same expression as ax8_term (ifa_gemv.hip)
    __device__ __forceinline__ float dot(const XRegsQ4<NJ> &X) const
    {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            int d = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint32_t cw = c[j][w];
                d = sdot4((int)(cw & 0x0F0F0F0Fu), X.xe[j][w], d);
                d = sdot4((int)((cw >> 4) & 0x0F0F0F0Fu), X.xo[j][w], d);
            }
            const float base = hbits2f((uint16_t)(sb[j] & 0xFFFFu));
            const float scale = hbits2f((uint16_t)(sb[j] >> 16));
            float t = (float)d * scale;
            float u = X.xsf[j] * base;
            t = t + u;
            acc = acc + X.xs[j] * t;
        }
        return acc;
    }
};

// ------------------------------------------------------------- kernel params
enum DecEpilogue { EPI_PLAIN = 0, EPI_RESIDUAL = 1, EPI_GLU = 2, EPI_ACT = 3 };

struct DecMatSet {
    const uint8_t *W[2];     // tiled rows; W[1] only for EPI_GLU (w3)
    const half_t *bias[2];
    half_t *y;
    int rows;
};

struct DecGemvParams {
    const half_t *x;           // activation [cols]
    const half_t *norm_w, *norm_b;
    float multi_base, eps;
    int cols, nblk;
    DecMatSet set[3];          // grid.y selects
    const half_t *residual;    // EPI_RESIDUAL: y = half(residual + y)
    const half_t *residual2;   // optional second add (parallel-attn / shared-input models)
    int act_kind;
    half_t *xn_out;            // optional copy of the normalised activation
    int rows_per_wave;
};

__device__ __forceinline__ half_t dec_bias(float acc, const half_t *bias, int row)
{
    half_t y = f2h(acc);
    if (bias) y = f2h(h2f(y) + h2f(bias[row]));
    return y;
}

template <int NJ, int R, int EPI, int NORM>
__global__ void __launch_bounds__(DEC_THREADS) k_dec_gemv_q4(const DecGemvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const XLds L = xlds_carve(smem, P.cols);
    const DecMatSet &S = P.set[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (DEC_THREADS / 64) + (threadIdx.x >> 6);
    const int row0 = gw * P.rows_per_wave;
    const int row_end = min(row0 + P.rows_per_wave, S.rows);
    const size_t row_bytes = (size_t)P.nblk * 20;
    constexpr int NM = (EPI == EPI_GLU) ? 2 : 1;

    // 1) first batch of weight loads, before anything that needs the activation
    WRowQ4<NJ> cur[NM][R];
#pragma unroll
    for (int m = 0; m < NM; m++)
#pragma unroll
        for (int rr = 0; rr < R; rr++)
            cur[m][rr].load(S.W[m] + (size_t)(row0 + rr) * row_bytes, P.nblk, lane, row0 + rr < row_end);

    // 2) norm + quantise the activation into LDS (all threads), then this lane's slice into registers
    dec_prologue<NORM>(P.x, P.norm_w, P.norm_b, P.multi_base, P.eps, P.cols, L, P.xn_out);
    if (row0 >= S.rows) return;
    XRegsQ4<NJ> X;
    X.load(L, lane, P.nblk);

    // 3) stream rows, next batch in flight while the current one is reduced
    for (int r = row0; r < row_end; r += R) {
        WRowQ4<NJ> nxt[NM][R];
        const bool more = r + R < row_end;
#pragma unroll
        for (int m = 0; m < NM; m++)
#pragma unroll
            for (int rr = 0; rr < R; rr++)
                nxt[m][rr].load(S.W[m] + (size_t)(r + R + rr) * row_bytes, P.nblk, lane, more && (r + R + rr < row_end));
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const int row = r + rr;
            float a0 = wave_sum(cur[0][rr].dot(X));
            float a1 = 0.0f;
            if constexpr (NM == 2) a1 = wave_sum(cur[1][rr].dot(X));
            if (lane == 0 && row < row_end) {
                half_t y = dec_bias(a0, S.bias[0], row);
                if constexpr (EPI == EPI_RESIDUAL) {
                    y = f2h(h2f(P.residual[row]) + h2f(y));             // TensorOpr::Add (half add)
                    if (P.residual2) y = f2h(h2f(y) + h2f(P.residual2[row]));
                } else if constexpr (EPI == EPI_GLU) {
                    half_t t2 = dec_bias(a1, S.bias[1], row);
                    half_t act = f2h(act_fn(h2f(y), P.act_kind));       // TensorOpr::Activation -> F16
                    y = f2h(h2f(act) * h2f(t2));                        // TensorOpr::Mul
                } else if constexpr (EPI == EPI_ACT) {
                    y = f2h(act_fn(h2f(y), P.act_kind));
                }
                S.y[row] = y;
            }
        }
#pragma unroll
        for (int m = 0; m < NM; m++)
#pragma unroll
            for (int rr = 0; rr < R; rr++) cur[m][rr] = nxt[m][rr];
    }
}

// ------------------------------------------------- final norm + F16 lm_head
struct DecLmHeadParams {
    const half_t *x;
    const half_t *norm_w, *norm_b;
    float multi_base, eps;
    int cols;
    const half_t *W;      // [rows][cols] F16
    half_t *logits;       // [rows]
    int rows;
    half_t *xn_out;
    int rows_per_wave;
};

// x chunk (8 halfs) per lane per j, activation normalised in the prologue and
// kept as F16 (the reference feeds the F16 norm output to GemvHalf_AX_Alg3).
template <int NJ, int R, int NORM>
__global__ void __launch_bounds__(DEC_THREADS) k_dec_lmhead_f16(const DecLmHeadParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *xn = reinterpret_cast<half_t *>(smem);                         // [cols]
    float *part = reinterpret_cast<float *>(smem + (((size_t)P.cols * 2 + 15) & ~(size_t)15));
    const int tid = threadIdx.x, lane = tid & 63;
    const int gw = blockIdx.x * (DEC_THREADS / 64) + (tid >> 6);
    const int row0 = gw * P.rows_per_wave;
    const int row_end = min(row0 + P.rows_per_wave, P.rows);
    const int chunks = P.cols >> 3;

    u32x4 cur[R][NJ];
#pragma unroll
    for (int rr = 0; rr < R; rr++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int c = lane + 64 * j;
            cur[rr][j] = u32x4{0, 0, 0, 0};
            if (row0 + rr < row_end && c < chunks)
                cur[rr][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(P.W + (size_t)(row0 + rr) * P.cols) + c);
        }

    float scale = 1.0f;
    if constexpr (NORM == 1) {
        if (tid < 128) part[tid] = rms_partial(P.x, P.cols, tid, 128);
        __syncthreads();
        if (tid == 0) part[128] = rms_scale_from_partials(part, 128, P.cols, P.eps);
        __syncthreads();
        scale = part[128];
    }
    for (int c = tid; c < chunks; c += DEC_THREADS) {
        half8_t xv = *reinterpret_cast<const half8_t *>(P.x + (size_t)c * 8);
        if constexpr (NORM == 1) {
            half8_t wv, bv;
            if (P.norm_w) wv = *reinterpret_cast<const half8_t *>(P.norm_w + (size_t)c * 8);
            if (P.norm_b) bv = *reinterpret_cast<const half8_t *>(P.norm_b + (size_t)c * 8);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float t = (float)xv[i] * scale;
                if (P.norm_w) {
                    float m = P.multi_base + (float)wv[i];
                    t = t * m;
                    if (P.norm_b) t = t + (float)bv[i];
                }
                xv[i] = f2h(t);
            }
        }
        *reinterpret_cast<half8_t *>(xn + (size_t)c * 8) = xv;
        if (P.xn_out && blockIdx.x == 0) *reinterpret_cast<half8_t *>(P.xn_out + (size_t)c * 8) = xv;
    }
    __syncthreads();
    if (row0 >= P.rows) return;
    u32x4 xr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int c = lane + 64 * j;
        xr[j] = u32x4{0, 0, 0, 0};
        if (c < chunks) xr[j] = *reinterpret_cast<const u32x4 *>(xn + (size_t)c * 8);
    }
    for (int r = row0; r < row_end; r += R) {
        u32x4 nxt[R][NJ];
        const bool more = r + R < row_end;
#pragma unroll
        for (int rr = 0; rr < R; rr++)
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int c = lane + 64 * j;
                nxt[rr][j] = u32x4{0, 0, 0, 0};
                if (more && r + R + rr < row_end && c < chunks)
                    nxt[rr][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(P.W + (size_t)(r + R + rr) * P.cols) + c);
            }
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; j++) acc = dot8_f16(cur[rr][j], xr[j], acc);
            acc = wave_sum(acc);
            if (lane == 0 && r + rr < row_end) P.logits[r + rr] = f2h(acc);
        }
#pragma unroll
        for (int rr = 0; rr < R; rr++)
#pragma unroll
            for (int j = 0; j < NJ; j++) cur[rr][j] = nxt[rr][j];
    }
}

// ------------------------------------------------------------------ attention
struct DecAttnParams {
    half_t *q;                 // [heads*head_dim]   (RoPE applied in LDS, not written back)
    const half_t *k_new;       // [kv_heads*head_dim] pre-RoPE
    const half_t *v_new;       // [kv_heads*head_dim]
    uint8_t *kcache, *vcache;  // [max_ctx][kv_row_bytes]
    const int *state;          // state[1] = position of the new token
    int heads, kv_heads, head_dim, kv_q8;
    float kq_scale, rope_theta;
    int rope_order, rope_dims, rope_cols;
    int alibi, alibi_base, alibi_total;
    half_t *out;               // [heads*head_dim]
    int max_ctx;
};

// One workgroup (256 threads) per query head.  Scores for the cached rows are
// computed one key per lane with the reference's k-ordered fp32 dot (bit-exact
// S), the new token's K/V come straight from registers/LDS (and are written to
// the cache by the first head of each KV group), softmax in LDS, then P.V with
// the context split over 256/head_dim thread groups.
template <bool Q8>
__global__ void __launch_bounds__(256) k_dec_attn(const DecAttnParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int hd = P.head_dim;
    const int pos = P.state[1];
    const int n_ctx = pos + 1;
    half_t *qs = reinterpret_cast<half_t *>(smem);                 // [hd] rotated q
    half_t *kn = qs + hd;                                          // [hd] rotated (and Q8 round-tripped) new k
    half_t *vn = kn + hd;                                          // [hd] new v (Q8 round-tripped)
    float *red = reinterpret_cast<float *>(vn + hd);               // [16]
    float *opart = red + 16;                                       // [256]
    half_t *S = reinterpret_cast<half_t *>(opart + 256);           // [n_ctx]
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = P.heads / P.kv_heads;
    const int kvh = h / group;
    const bool writer = (h % group) == 0;
    const int kv_dim = P.kv_heads * hd;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;

    // ---- stage q, k_new, v_new; RoPE on q and k (TensorOpr::PositionEmbedding, F16 in/out)
    for (int d = tid; d < hd; d += 256) {
        qs[d] = P.q[(size_t)h * hd + d];
        kn[d] = P.k_new[(size_t)kvh * hd + d];
        vn[d] = P.v_new[(size_t)kvh * hd + d];
    }
    __syncthreads();
    if (P.rope_order != 0) {
        for (int c = tid; c < hd; c += 256) {     // first hd/2 threads rotate q, next hd/2 rotate k
            if (c < hd / 2) rope_rotate(qs, c, pos, P.rope_theta, P.rope_order, P.rope_dims, P.rope_cols);
            else rope_rotate(kn, c - hd / 2, pos, P.rope_theta, P.rope_order, P.rope_dims, P.rope_cols);
        }
        __syncthreads();
    }
    // ---- KV store of the new row (LayerKVCache::SetKRows/SetVRows, kv_cache.cc:159-249)
    if constexpr (Q8) {
        // quantise 32-element blocks of this head's slice (head_dim % 32 == 0), round-trip for local use
        const int nb = hd / 32;
        for (int b = wave; b < 2 * nb; b += 4) {
            half_t *src = b < nb ? kn : vn;
            const int bb = b < nb ? b : b - nb;
            if (lane < 32) {
                const float val = h2f(src[bb * 32 + lane]);
                float mx = fabsf(val);
#pragma unroll
                for (int m = 16; m > 0; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 32));
                const float sc = mx / 127;
                int qv = sc <= 0.000001f ? 0 : (int)roundf(val / sc);
                qv = min(max(qv, -128), 127);
                const half_t sch = f2h(sc);
                if (writer) {
                    uint8_t *cache = b < nb ? P.kcache : P.vcache;
                    uint8_t *blk = cache + (size_t)pos * row_bytes + (size_t)((kvh * hd) / 32 + bb) * 34;
                    blk[2 + lane] = (uint8_t)(int8_t)qv;
                    if (lane == 0) *reinterpret_cast<uint16_t *>(blk) = __builtin_bit_cast(uint16_t, sch);
                }
                src[bb * 32 + lane] = f2h((float)qv * h2f(sch));   // dequantised value, as GetKRows returns it
            }
        }
        __syncthreads();
    } else {
        if (writer) {
            half_t *kc = reinterpret_cast<half_t *>(P.kcache + (size_t)pos * row_bytes) + (size_t)kvh * hd;
            half_t *vc = reinterpret_cast<half_t *>(P.vcache + (size_t)pos * row_bytes) + (size_t)kvh * hd;
            for (int d = tid; d < hd; d += 256) { kc[d] = kn[d]; vc[d] = vn[d]; }
        }
    }

    // ---- scores: one key per lane, fp32 fma in d order (Gemm_Alg2_Kernel order, products exact)
    const float alpha = 1.0f / sqrtf((float)hd) / P.kq_scale;
    const float mk = P.alibi ? alibi_slope(h + P.alibi_base, P.alibi_total) : 0.0f;
    float lmax = -INFINITY;
    for (int j = tid; j < n_ctx; j += 256) {
        float c = 0.0f;
        if (j == pos) {
            for (int d = 0; d < hd; d++) c = __builtin_fmaf(h2f(qs[d]), h2f(kn[d]), c);
        } else if constexpr (Q8) {
            const uint8_t *rowp = P.kcache + (size_t)j * row_bytes + (size_t)((kvh * hd) / 32) * 34;
            for (int b = 0; b < hd / 32; b++) {
                const uint8_t *blk = rowp + (size_t)b * 34;
                const float sc = hbits2f(*reinterpret_cast<const uint16_t *>(blk));
#pragma unroll 8
                for (int i = 0; i < 32; i++) {
                    const float kvv = h2f(f2h((float)(int)(int8_t)blk[2 + i] * sc));
                    c = __builtin_fmaf(h2f(qs[b * 32 + i]), kvv, c);
                }
            }
        } else {
            const half8_t *rowp = reinterpret_cast<const half8_t *>(P.kcache + (size_t)j * row_bytes) + (size_t)(kvh * hd) / 8;
            for (int d8 = 0; d8 < hd / 8; d8++) {
                const half8_t kv8 = rowp[d8];
#pragma unroll
                for (int i = 0; i < 8; i++) c = __builtin_fmaf(h2f(qs[d8 * 8 + i]), (float)kv8[i], c);
            }
        }
        half_t s = f2h(alpha * c);
        if (P.alibi) { float a = (float)j * mk; s = f2h(a + h2f(s)); }
        S[j] = s;
        lmax = fmaxf(lmax, P.kq_scale * h2f(s));
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.0f;
    for (int j = tid; j < n_ctx; j += 256) {
        const float e = expf(P.kq_scale * h2f(S[j]) - mx);
        lsum += e;
        S[j] = f2h(e);
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / (((red[4] + red[5]) + red[6]) + red[7]);
    for (int j = tid; j < n_ctx; j += 256) S[j] = f2h(h2f(S[j]) * inv);
    __syncthreads();

    // ---- O = P.V : thread (split, d); each split walks its slice of the context in order
    const int nsplit = 256 / hd > 0 ? 256 / hd : 1;
    const int d = tid % hd, sp = tid / hd;
    float c = 0.0f;
    if (sp < nsplit) {
        const int per = (n_ctx + nsplit - 1) / nsplit;
        const int j0 = sp * per, j1 = min(n_ctx, j0 + per);
        for (int j = j0; j < j1; j++) {
            float vv;
            if (j == pos) vv = h2f(vn[d]);
            else if constexpr (Q8) {
                const uint8_t *blk = P.vcache + (size_t)j * row_bytes + (size_t)((kvh * hd + d) / 32) * 34;
                vv = h2f(f2h((float)(int)(int8_t)blk[2 + ((kvh * hd + d) & 31)] * hbits2f(*reinterpret_cast<const uint16_t *>(blk))));
            } else {
                vv = h2f(reinterpret_cast<const half_t *>(P.vcache + (size_t)j * row_bytes)[(size_t)kvh * hd + d]);
            }
            c = __builtin_fmaf(h2f(S[j]), vv, c);
        }
    }
    opart[tid] = c;
    __syncthreads();
    if (tid < hd) {
        float o = opart[tid];
        for (int s2 = 1; s2 < nsplit; s2++) o = o + opart[s2 * hd + tid];
        P.out[(size_t)h * hd + tid] = f2h(o);
    }
}

__host__ __device__ inline size_t dec_attn_smem(int head_dim, int max_ctx)
{
    return (size_t)head_dim * 3 * 2 + 16 * 4 + 256 * 4 + (((size_t)max_ctx * 2 + 15) & ~(size_t)15) + 16;
}

// ------------------------------------------------------------- small kernels
// state[0] = current token id, state[1] = its position, state[2] = steps done;
// state[8 + i] = i-th generated token of the current launch batch.
__global__ void __launch_bounds__(256) k_dec_gather(const half_t *__restrict__ embd, const int *__restrict__ state,
                                                    int dim, int vocab, half_t *__restrict__ x)
{
    int tok = state[0];
    tok = min(max(tok, 0), vocab - 1);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < dim / 8; c += gridDim.x * blockDim.x)
        reinterpret_cast<u32x4 *>(x)[c] = reinterpret_cast<const u32x4 *>(embd + (size_t)tok * dim)[c];
}

// greedy top-1 over the logits (first maximum wins); writes the token ring and advances the state
__global__ void __launch_bounds__(1024) k_dec_argmax_advance(const half_t *__restrict__ v, int n, int *__restrict__ state,
                                                             int ring)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY; int besti = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float f = h2f(v[i]);
        if (f > best || (f == best && i < besti)) { best = f; besti = i; }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        float ob = __shfl_xor(best, m); int oi = __shfl_xor(besti, m);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        if (besti == 0x7FFFFFFF) besti = 0;
        const int step = state[2];
        state[8 + (step % ring)] = besti;
        state[0] = besti;
        state[1] = state[1] + 1;
        state[2] = step + 1;
    }
}

} // namespace ifa
