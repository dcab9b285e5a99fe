// ifa_decode_kernels.h -- fused batch-1 decode kernels (the hot path).
//
// One decoder layer at T=1 is 5 launches instead of the reference's ~30
// launch+sync pairs:
//   k_dec_gemv<QKV>   : [RMSNorm -> Q8 act-quant] prologue + Wq/Wk/Wv GEMV (+bias)
//   k_dec_attn        : RoPE(q,k) + KV-cache write (F16 or Q8) + GQA attention
//   k_dec_gemv<WO>    : [Q8 act-quant] + Wo GEMV + bias + residual add
//   k_dec_gemv<FFN13> : [RMSNorm -> quant] + W1,W3 GEMV + act(t1)*t2
//   k_dec_gemv<W2>    : [quant] + W2 GEMV + bias + residual add(s)
// plus embedding gather (+ the step's RoPE table), final-norm + F16 lm_head GEMV
// and a device-side greedy argmax that also advances the position, so a whole
// step is graph-replayable.
//
// Rounding points are the reference's (every op boundary is an F16 tensor); the
// prologue reproduces Tensor_RmsNorm_Kernel's partial-sum order exactly, so the
// int8 activation codes are bit-identical to the op-by-op path.
//
// Weight rows are streamed from the row-local plane layout (ifa_tiled.h):
// lane l owns blocks l+64j of every row, issues one aligned 16-byte load per
// block plus a 4-byte (base,scale) load, and keeps its slice of the int8
// activation in registers for all rows it processes.  Each kernel runs ONE
// resident wave set (<= 2 workgroups per CU); a wave walks row batches
// b = wave, wave+W, ... so the prologue is paid once, the first batch of weight
// loads is issued BEFORE the prologue (HBM latency overlaps norm/quant) and the
// next batch is always in flight while the current one is reduced.
#pragma once
#include "ifa_device.h"
#include "ifa_math.h"
#include "ifa_tiled.h"
#include "ifa_decode_formats.h"

namespace ifa {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// q = roundf(v / qs) exactly as Tensor_QuantizeQ8_B32T2_Alg2_Kernel computes it (src/kernels/tensor_quant.h:61-81),
// without paying an IEEE division per element: t = v * (1/qs) is within 1.5 ulp of the exact quotient and the fp32
// quotient within 0.5 ulp, |v/qs| <= 127, so the two are less than 2^-15 apart and round-to-nearest of t equals
// roundf(v/qs) unless t sits within 2^-14 of a half-integer; then the true division decides.  The slow path is ONE
// copy of the eight divisions behind a flag (it is taken for ~0.1 % of the chunks): the previous form -- a branch with an
// inlined division per element -- made the single-shot prologue the bulk of the kernel's code (40 divisions, 112 exec
// branches in the Wo kernel) and ran the division for every fifth wave-instruction.
__device__ __forceinline__ void q8_round_div8(const float (&v)[8], float qs, int (&q)[8])
{
    const float rqs = 1.0f / qs;
    bool risky = false;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float t = v[i] * rqs;
        const float k = __builtin_rintf(t);
        risky |= fabsf(fabsf(t - k) - 0.5f) < 0.00006103515625f;
        q[i] = (int)k;
    }
    if (__builtin_expect(risky, 0)) {
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = (int)roundf(v[i] / qs);
    }
    if (qs <= 0.000001f) {
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = min(max(q[i], -128), 127);
}

// one value (attention epilogue: a lane per element)
__device__ __forceinline__ int q8_round_div1(float v, float qs)
{
    if (qs <= 0.000001f) return 0;
    const float t = v * (1.0f / qs);
    float k = __builtin_rintf(t);
    if (__builtin_expect(fabsf(fabsf(t - k) - 0.5f) < 0.00006103515625f, 0)) k = roundf(v / qs);
    return min(max((int)k, -128), 127);
}

// Four Q8_B32T2 codes (one dword of int8) times their block's F16 scale -> four halves, each EXACTLY
// half((float)q * scale), the value LayerKVCache::GetKRows / GetVRows hand back (kv_cache.cc:104-157): q in [-128, 127] is
// exact in half -- built without a conversion as (1024 + (q ^ 0x80)) - 1152: the half with bits 0x6400 | u is 1024 + u --
// and the product of two halves is exact in fp32, so ONE packed half multiply (round to nearest even) rounds the same
// exact value the reference rounds.  3 packed instructions per 2 values instead of sign-extend / cvt / mul / cvt per value:
// the Q8 cache had cost the decode attention 2.3 us per layer over the F16 one (9.8 vs 7.5 us).
__device__ __forceinline__ void q8x4_dequant_h(uint32_t w, half2_t sc2, half2_t &lo, half2_t &hi)
{
    const uint32_t x = w ^ 0x80808080u;
    const uint32_t a = __builtin_amdgcn_perm(0x64646464u, x, 0x04010400u);      // halves 1024 + u0, 1024 + u1
    const uint32_t b = __builtin_amdgcn_perm(0x64646464u, x, 0x04030402u);      // halves 1024 + u2, 1024 + u3
    const half2_t bias = {(half_t)-1152.0f, (half_t)-1152.0f};
    lo = (__builtin_bit_cast(half2_t, a) + bias) * sc2;
    hi = (__builtin_bit_cast(half2_t, b) + bias) * sc2;
}

#ifndef IFA_NT_WEIGHTS
#define IFA_NT_WEIGHTS 1            // stream weights with the non-temporal policy (read once per token)
#endif
constexpr int DEC_THREADS = 512;   // 8 waves per workgroup
constexpr int DEC_WAVES = DEC_THREADS / 64;

// ------------------------------------------------------------------ LDS image
// of the (normalised and) quantised activation vector
struct XLds {
    int8_t *codes;   // [cols], natural element order, 16-byte aligned
    float *scale;    // [cols/32]  fp32 value of the fp16-rounded block scale
    float *xsum;     // [cols/32]  sum of the 32 int8 codes (exact in fp32)
    float *part;     // [128] partial sums + [4] stats
    half_t *xh;      // [cols] staged input
};

__host__ __device__ inline size_t xlds_bytes(int cols)
{
    size_t nb = (size_t)cols / 32;
    size_t off = ((size_t)cols + 15) / 16 * 16 + nb * 8;
    off = (off + 15) / 16 * 16;
    off += 132 * 4;
    off = (off + 15) / 16 * 16;
    off += (size_t)cols * 2;
    return off + 16;
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
    l.part = reinterpret_cast<float *>(smem + off); off += 132 * 4;
    off = (off + 15) / 16 * 16;
    l.xh = reinterpret_cast<half_t *>(smem + off);
    return l;
}

// Global image of a quantised activation left by the kernel that PRODUCED it (k_dec_attn for Wo): the same three
// arrays as the LDS image -- int8 codes [cols], fp32 value of the fp16-rounded block scale [cols/32], code sums
// [cols/32] -- so the consuming GEMV starts streaming at once instead of re-quantising the vector on every CU.
struct XqImage {
    int8_t *codes; float *scale; float *xsum;
};
__host__ __device__ inline size_t xq_image_bytes(int cols) { return (((size_t)cols + 15) / 16 * 16) + ((size_t)cols / 32) * 8 + 16; }
__host__ __device__ inline XqImage xq_image_carve(void *base, int cols)
{
    XqImage q;
    q.codes = reinterpret_cast<int8_t *>(base);
    q.scale = reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (((size_t)cols + 15) / 16 * 16));
    q.xsum = q.scale + cols / 32;
    return q;
}

// Prologue shared by every decode GEMV kernel.  NORM: 0 none, 1 RMS (2: no prologue at all, the activation arrives
// quantised in an XqImage).
//   xn = NORM ? half(rms(x)) : x ;  Q8_B32T2 quantisation of xn exactly as
//   Tensor_QuantizeQ8_B32T2_Alg2_Kernel (src/kernels/tensor_quant.h:44-82).
// Split in two so that the activation loads are the FIRST memory operations of
// the kernel (loads return in issue order per wave: issued after the weight
// stream they would only arrive once that stream has drained).
//   XPre pre; pre.issue(...);   ... issue weight loads ...;   pre.finish(...);
// Must be executed by all threads of the workgroup.  cols % 32 == 0, cols <= 8*blockDim*MAXC.
// XADD: the activation is the sum x + (add [+ add_bias]) of two vectors (tensor parallelism: layer input + the
// all-reduced product, bias once after the merge); the sum -- two half additions in TensorOpr::Add order -- replaces x
// and workgroup 0 stores it for the residual that follows.
// NT > 0 (wave-specialised kernels, k_dec_gemv<.., NP>): only threads [0, NT) -- the workgroup's first NT / 64 waves -- run the
// prologue; they synchronise with each other through two LDS counters (part[130]: partial sums written, part[131]: codes
// written) instead of s_barrier, so that the OTHER waves can sit behind their weight requests without holding the prologue
// up, and everybody waits for part[131] == NT / 64 before reading the image (xpre_wait_image).  Same chunk -> lane / group
// assignment rule (chunk c = tid + k * threads: lane c % 64 of group c / 64), so the statistics and the codes are the
// same bits for any NT.
__device__ __forceinline__ void lds_counter_add(float *slot)
{
    asm volatile("" ::: "memory");      // the wave's earlier LDS writes stay ahead of the counter (the LDS performs a wave's accesses in order)
    __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(slot), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_counter_wait(float *slot, uint32_t target)
{
    volatile uint32_t *c = reinterpret_cast<volatile uint32_t *>(slot);
    while (__builtin_amdgcn_readfirstlane(*c) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

template <int NORM, int MAXC, bool XADD = false, int NT = 0>
struct XPre {
    static __device__ __forceinline__ int nthr() { return NT ? NT : (int)blockDim.x; }
    half8_t xv[MAXC];
    half8_t wv[NORM ? MAXC : 1], bv[NORM ? MAXC : 1];
    half8_t av[XADD ? MAXC : 1], abv[XADD ? MAXC : 1];

    __device__ __forceinline__ void issue_add(const half_t *__restrict__ add, const half_t *__restrict__ add_bias, int cols)
    {
        const int chunks = cols >> 3;
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            const int c = threadIdx.x + k * (int)blockDim.x;
            if (c < chunks) {
                av[k] = *reinterpret_cast<const half8_t *>(add + (size_t)c * 8);
                if (add_bias) abv[k] = *reinterpret_cast<const half8_t *>(add_bias + (size_t)c * 8);
            }
        }
    }
    __device__ __forceinline__ void apply_add(const half_t *__restrict__ add_bias, int cols, half_t *__restrict__ sum_out)
    {
        const int chunks = cols >> 3;
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            const int c = threadIdx.x + k * (int)blockDim.x;
            if (c >= chunks) continue;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                half_t t = av[k][i];
                if (add_bias) t = f2h(h2f(t) + h2f(abv[k][i]));       // Add(reduced, bias)
                xv[k][i] = f2h(h2f(xv[k][i]) + h2f(t));               // Add(layer_input, .)
            }
            if (sum_out && blockIdx.x == 0) *reinterpret_cast<half8_t *>(sum_out + (size_t)c * 8) = xv[k];
        }
    }

    __device__ __forceinline__ void issue(const half_t *__restrict__ x, const half_t *__restrict__ nw,
                                          const half_t *__restrict__ nb, int cols)
    {
        const int chunks = cols >> 3;
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            const int c = threadIdx.x + k * nthr();
            if (c < chunks) {
                xv[k] = *reinterpret_cast<const half8_t *>(x + (size_t)c * 8);
                if constexpr (NORM == 1) {
                    if (nw) wv[k] = *reinterpret_cast<const half8_t *>(nw + (size_t)c * 8);
                    if (nb) bv[k] = *reinterpret_cast<const half8_t *>(nb + (size_t)c * 8);
                }
            }
        }
    }

    // the norm weights alone (ifa_decode_chain.h: the values arrive as granules, gathered into xv by the caller)
    __device__ __forceinline__ void issue_norm(const half_t *__restrict__ nw, const half_t *__restrict__ nb, int cols)
    {
        if constexpr (NORM == 1) {
            const int chunks = cols >> 3;
#pragma unroll
            for (int k = 0; k < MAXC; k++) {
                const int c = threadIdx.x + k * nthr();
                if (c < chunks) {
                    if (nw) wv[k] = *reinterpret_cast<const half8_t *>(nw + (size_t)c * 8);
                    if (nb) bv[k] = *reinterpret_cast<const half8_t *>(nb + (size_t)c * 8);
                }
            }
        }
    }

    __device__ __forceinline__ void finish(const half_t *__restrict__ nw, const half_t *__restrict__ nb,
                                           float multi_base, float eps, int cols, const XLds &L,
                                           half_t *__restrict__ xn_out, long long *trc = nullptr)
    {
        const int tid = threadIdx.x;
        const int chunks = cols >> 3;
        float scale = 1.0f;
        if constexpr (NORM == 1) {
            // sum of squares in the canonical order of ifa_math.h: chunk c = tid + k*blockDim is lane c % 64 of group
            // c / 64 = wave + k * (blockDim / 64); no staging of x, one barrier
            const int lane = tid & 63, wave = tid >> 6, nwaves = nthr() >> 6;
#pragma unroll
            for (int k = 0; k < MAXC; k++) {
                const int c = tid + k * nthr();
                half8_t v8 = xv[k];
                if (c >= chunks) {
#pragma unroll
                    for (int i = 0; i < 8; i++) v8[i] = (half_t)0;
                }
                const float pg = wave_sum(rms_chunk_sq(v8));
                if (lane == 0) L.part[wave + k * nwaves] = pg;
            }
            if constexpr (NT > 0) { if (lane == 0) lds_counter_add(L.part + 130); lds_counter_wait(L.part + 130, NT / 64); }
            else __syncthreads();
            if (trc) trc[5] = wall_clock64();
            // the group sums in ascending order (rms_total's order), read as 16-byte words with the groups past the row end
            // masked to +0 (exact): a loop over the runtime group count was one dependent LDS round trip per group
            const int ng = (chunks + 63) >> 6;
            float total = 0.0f;
            typedef float f4p __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int v4 = 0; v4 < 4 * MAXC; v4++) {          // <= 16 waves x MAXC chunks per thread
                const f4p p = reinterpret_cast<const f4p *>(L.part)[v4];
#pragma unroll
                for (int i = 0; i < 4; i++) total = total + ((v4 * 4 + i) < ng ? p[i] : 0.0f);
            }
            scale = rms_scale_of(total, cols, eps);
            if (trc) trc[6] = wall_clock64();
        } else {
            if (trc) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trc[4] = wall_clock64(); }
        }
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            const int c = tid + k * nthr();
            if (c >= chunks) continue;       // whole quads (4 lanes = one block) are in or out together
            float v[8];
            if constexpr (NORM == 1) {
                half8_t outv;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float t = (float)xv[k][i] * scale;
                    if (nw) {
                        float m = multi_base + (float)wv[k][i];
                        t = t * m;
                        if (nb) t = t + (float)bv[k][i];
                    }
                    half_t th = f2h(t);
                    outv[i] = th;
                    v[i] = h2f(th);
                }
                if (xn_out && blockIdx.x == 0) *reinterpret_cast<half8_t *>(xn_out + (size_t)c * 8) = outv;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = (float)xv[k][i];
            }
            float mx = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
            mx = fmaxf(mx, dpp_xor1(mx));
            mx = fmaxf(mx, dpp_xor2(mx));
            const float qs = mx / 127;
            int q[8]; int s = 0;
            q8_round_div8(v, qs, q);
#pragma unroll
            for (int i = 0; i < 8; i++) s += q[i];
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
        if constexpr (NT > 0) { if ((tid & 63) == 0) lds_counter_add(L.part + 131); }
        else __syncthreads();
    }
};

// ------------------------------------------------------------------ Q4_B32T1
// registers of one lane for NJ blocks of the activation / of one weight row
template <int NJ>
struct XRegsQ4 {
    int xe[NJ][4], xo[NJ][4];
    float xs[NJ], xsf[NJ];
    // unconditional reads with the block index clamped (LDS image, or the global image a producing kernel left: loads
    // under an exec mask would make every later vmcnt wait a vmcnt(0)); blocks past the end are zeroed by selects
    __device__ __forceinline__ void load(const int8_t *codes, const float *scale, const float *xsum, int lane, int nblk, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = blk0 + lane + 64 * j;
            const bool ok = blk < nblk;
            const int cb = ok ? blk : nblk - 1;
            const u32x4 a = *reinterpret_cast<const u32x4 *>(codes + (size_t)cb * 32);
            const u32x4 b = *reinterpret_cast<const u32x4 *>(codes + (size_t)cb * 32 + 16);
            const float sc = scale[cb], su = xsum[cb];
            const uint32_t d[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int e = (int)__builtin_amdgcn_perm(d[2 * w + 1], d[2 * w], 0x06040200u);
                const int o = (int)__builtin_amdgcn_perm(d[2 * w + 1], d[2 * w], 0x07050301u);
                xe[j][w] = ok ? e : 0; xo[j][w] = ok ? o : 0;
            }
            xs[j] = ok ? sc : 0.0f;
            xsf[j] = ok ? su : 0.0f;
        }
    }
};

template <int NJ>
struct WRowQ4 {
    u32x4 c[NJ];
    uint32_t sb[NJ];
    // Unconditional loads (addresses clamped into the row): a load under an exec-masked
    // branch makes the compiler's vmcnt bookkeeping conservative -- every later wait
    // for OLDER data (e.g. the activation) turns into vmcnt(0), i.e. "wait for the whole
    // weight stream".  Lanes past the row end re-read the last block; their activation
    // registers are zero, so they contribute exactly 0.
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(blk0 + lane + 64 * j, nblk - 1);
#if IFA_NT_WEIGHTS
            c[j] = nt_load<u32x4>(wrow + (size_t)blk * 16);              // (global address space stated: see nt_load)
            sb[j] = nt_load<uint32_t>(wrow + (size_t)nblk * 16 + (size_t)blk * 4);
#else
            c[j] = *reinterpret_cast<const u32x4 *>(wrow + (size_t)blk * 16);
            sb[j] = *reinterpret_cast<const uint32_t *>(wrow + (size_t)nblk * 16 + (size_t)blk * 4);
#endif
        }
    }
    template <class S>
    __device__ __forceinline__ void load_src(const S &src, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint32_t blk = (uint32_t)min(blk0 + lane + 64 * j, nblk - 1);
            c[j] = src.template ld<u32x4>(blk * 16);
            sb[j] = src.template ld<uint32_t>((uint32_t)nblk * 16 + blk * 4);
        }
    }
    // lane-partial of sum_blk xs*(dot*scale + xsum*base); same expression as ax8_term (ifa_gemv.hip)
    __device__ __forceinline__ float dot(const XRegsQ4<NJ> &X, float acc0 = 0.0f) const
    {
        float acc = acc0;
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

// format -> register images.  DW = VGPRs one block of one row costs a lane (sizes the rows in flight).
template <int DT, int NJ> struct DecFmt;
template <int NJ> struct DecFmt<Q4_B32T1A, NJ> { using X = XRegsQ4<NJ>; using W = WRowQ4<NJ>; static constexpr int DW = 5; };
template <int NJ> struct DecFmt<Q8_B32T2, NJ> { using X = XRegsNat<NJ>; using W = WRowQ8T2<NJ>; static constexpr int DW = 9; };
template <int NJ> struct DecFmt<Q4_B64T1, NJ> { using X = XRegsB64<NJ>; using W = WRowQ4B64<NJ>; static constexpr int DW = 9; };
template <int NJ> struct DecFmt<Q3H_B64T1, NJ> { using X = XRegsB64<NJ>; using W = WRowQ4B64<NJ>; static constexpr int DW = 9; };      // nibble-pair tiled form (ifa_tiled.h)
template <int NJ> struct DecFmt<Q3H_NATIVE, NJ> { using X = XRegsB64<NJ>; using W = WRowQ3HN<NJ>; static constexpr int DW = 8; };       // native 32-byte form (option q3h_native)
template <int NJ> struct DecFmt<Q5_B64T1, NJ> { using X = XRegsB64<NJ>; using W = WRowQ5B64<NJ>; static constexpr int DW = 11; };
template <int NJ> struct DecFmt<Q6_B64T1, NJ> { using X = XRegsB64<NJ>; using W = WRowQ6B64<NJ>; static constexpr int DW = 13; };

// ------------------------------------------------------------- kernel params
// EPI_MOE_ACC / EPI_MOE_LAST: y = hfma(product, w_expert, y) (AddByRowIdx_Kernel); LAST also adds the residual(s)
// EPI_MOE_GLU / EPI_MOE_ACT: EPI_GLU / EPI_ACT of an expert.  Only the MOE epilogues contain the table lookup: a
// run-time `if (P.w_table)` in the dense kernels cost them ~2 us each (weight pointers forced through VGPRs).
enum DecEpilogue { EPI_PLAIN = 0, EPI_RESIDUAL = 1, EPI_GLU = 2, EPI_ACT = 3, EPI_MOE_ACC = 4, EPI_MOE_LAST = 5, EPI_MOE_GLU = 6, EPI_MOE_ACT = 7 };
constexpr bool epi_is_moe(int epi) { return epi >= EPI_MOE_ACC; }
constexpr bool epi_is_glu(int epi) { return epi == EPI_GLU || epi == EPI_MOE_GLU; }

// Kernel arguments.  Kept under 256 bytes with the fields the prologue needs first: when the struct grew past 256 B
// the kernels whose first instructions read the tail (xn_out, trace) started ~0.7 us later (second kernarg fetch).
// Up to three matrices ("sets") are concatenated into one virtual row space (Wq|Wk|Wv); EPI_GLU pairs W0[0] with W1.
struct DecGemvParams {
    const half_t *x;           // activation [cols] (NORM == 2: the XqImage the producing kernel left instead, see xq_image_carve)
    const half_t *norm_w, *norm_b;
    float multi_base, eps;
    int cols, nblk;            // nblk = WEIGHT blocks per row (cols / block capacity of the format)
    short nsets, act_kind;     // (16-bit fields: the struct is kept at 248 bytes, see below)
    int total_rows;
    const uint8_t *W0[3];      // tiled rows of each set
    const uint8_t *W1;         // w3 (EPI_GLU), paired with W0[0]
    int rows[3];
    float post_scale;          // see pre_scale
    half_t *xn_out;            // optional copy of the normalised activation
    long long *trace;          // optional [gridDim.x][8] wall-clock stamps (100 MHz) for tuning
    half_t *y[3];
    const half_t *b0[3];
    const half_t *b1;
    const half_t *residual;    // EPI_RESIDUAL: y = half(residual + y)
    const half_t *residual2;   // optional second add (parallel-attn / shared-input models)
    const half_t *x_add, *x_add_bias;   // XADD kernels: activation = x + (x_add [+ x_add_bias]), stored to xsum_out
    half_t *xsum_out;
    // mixture of experts: the weights of set 0 come from a device-side table indexed by the expert id the router
    // kernel chose for slot `moe_slot` (w_table[4*e + {0: w1, 1: w3, 2: w2}]); moe_w[slot] = its half weight
    const uint8_t *const *w_table;
    const int *moe_sel;
    const half_t *moe_w;
    const half_t *moe_acc;     // running sum over the experts visited so far (read when moe_slot > 0)
    short moe_slot, moe_tab_off;
    // TensorOpr::Scale of the reference's layer wiring, 0 = absent: pre_scale multiplies the product (+ bias) before the
    // residual is added (attn_out_scale / ffn_out_scale, inference_worker.cc:841-843, 927-929), post_scale the sum after it
    // (out_scale on the last layer's output); each is its own F16 rounding like the separate Scale launch it replaces
    float pre_scale;
};
// (tools/probes/gap_probe.hip: the boundary between two trivial launches is 1.58-1.60 us for any workgroup size, grid,
// LDS size and argument-block size from 64 to 512 bytes; what matters is which fields the first instructions wait for)
static_assert(sizeof(DecGemvParams) <= 248, "DecGemvParams grew: keep the fields the prologue reads first at the front");

__device__ __forceinline__ half_t dec_bias(float acc, const half_t *bias, int row)
{
    half_t y = f2h(acc);
    if (bias) y = f2h(h2f(y) + h2f(bias[row]));
    return y;
}

// virtual row -> (set, row).  Everything is selected from kernel-argument scalars:
// indexing P.set[] with a runtime index would make the compiler fetch the pointer
// from the kernarg segment with a VECTOR load and wait vmcnt(0) for it -- i.e. for
// every weight load issued before it -- which serialises the whole stream.
struct DecRow {
    int si, row;
    const uint8_t *W0, *W1;
    const half_t *b0, *b1;
    half_t *y;
};

__device__ __forceinline__ DecRow dec_locate(const DecGemvParams &P, int v)
{
    const int r0 = P.rows[0], r1 = P.rows[1];
    DecRow d;
    const bool in1 = P.nsets > 1 && v >= r0;
    const bool in2 = P.nsets > 2 && v >= r0 + r1;
    d.si = in2 ? 2 : (in1 ? 1 : 0);
    d.row = in2 ? v - r0 - r1 : (in1 ? v - r0 : v);
    d.W0 = in2 ? P.W0[2] : (in1 ? P.W0[1] : P.W0[0]);
    d.W1 = P.W1;
    d.b0 = in2 ? P.b0[2] : (in1 ? P.b0[1] : P.b0[0]);
    d.b1 = P.b1;
    d.y = in2 ? P.y[2] : (in1 ? P.y[1] : P.y[0]);
    return d;
}

// lane-local end of a row: bias, then the epilogue of the fused op sequence
// res / res2: P.residual[row] / P.residual2[row], requested right behind the row's weights (EPI_RESIDUAL)
template <int EPI>
__device__ __forceinline__ half_t dec_row_value(const DecGemvParams &P, const DecRow d, float a0, float a1, half_t res = (half_t)0,
                                                half_t res2 = (half_t)0)
{
    const int row = d.row;
    half_t y = dec_bias(a0, d.b0, row);
    if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_PLAIN) {
        if (P.pre_scale != 0.0f) y = f2h(h2f(y) * P.pre_scale);      // TensorOpr::Scale (k_scale)
    }
    if constexpr (EPI == EPI_RESIDUAL) {
        y = f2h(h2f(res) + h2f(y));                         // TensorOpr::Add (half add)
        if (P.residual2) y = f2h(h2f(y) + h2f(res2));
        if (P.post_scale != 0.0f) y = f2h(h2f(y) * P.post_scale);
    } else if constexpr (EPI == EPI_GLU || EPI == EPI_MOE_GLU) {
        half_t t2 = dec_bias(a1, d.b1, row);
        half_t act = f2h(act_fn(h2f(y), P.act_kind));       // TensorOpr::Activation -> F16
        y = f2h(h2f(act) * h2f(t2));                        // TensorOpr::Mul
    } else if constexpr (EPI == EPI_ACT || EPI == EPI_MOE_ACT) {
        y = f2h(act_fn(h2f(y), P.act_kind));
    } else if constexpr (EPI == EPI_MOE_ACC || EPI == EPI_MOE_LAST) {
        const half_t wexp = P.moe_w[P.moe_slot];
        const half_t prev = P.moe_slot == 0 ? (half_t)0 : P.moe_acc[row];
        y = __builtin_fmaf16(y, wexp, prev);                // TensorOpr::AddByRowIndex
        if constexpr (EPI == EPI_MOE_LAST) {
            if (P.pre_scale != 0.0f) y = f2h(h2f(y) * P.pre_scale);
            y = f2h(h2f(y) + h2f(P.residual[row]));         // + residual (Add(ff_out, residual))
            if (P.residual2) y = f2h(h2f(y) + h2f(P.residual2[row]));
            if (P.post_scale != 0.0f) y = f2h(h2f(y) * P.post_scale);
        }
    }
    return y;
}

template <int EPI>
__device__ __forceinline__ void dec_finish_row(const DecGemvParams &P, const DecRow d, float a0, float a1, half_t res = (half_t)0,
                                               half_t res2 = (half_t)0)
{
    d.y[d.row] = dec_row_value<EPI>(P, d, a0, a1, res, res2);
}

// RW = rows (EPI_GLU: row pairs) per wave and pass; rows are strided over the waves
// (v = (pass*RW + i)*W + wave) so neighbouring waves stream neighbouring rows.
// Order of memory traffic inside the kernel (measured, see DESIGN.md "Kernel timeline"):
//   1. activation (+norm weight) loads by all threads, then a barrier: the CU's memory
//      pipeline is FIFO across waves, so these must be queued before any weight request;
//   2. D1 rows of weights per wave (~40 KiB per CU: accepted without blocking);
//   3. cooperative norm + Q8 quantisation through LDS (VALU-bound, ~2 us);
//   4. the remaining rows -- all of them at once, nothing waits on them until the dots;
//   5. every wave reduces its RW rows as independent chains, then lane i finishes row i
//      (bias / residual / activation) so the epilogue's loads overlap too.
// TH = threads per workgroup: 512 (two waves per SIMD, 256 registers each) or 1024 (four waves per SIMD, 128 registers, half
// the rows in flight per wave) -- chosen per kernel shape by the launcher (ifa_decode_gemv_impl.h)
// NP > 0: WAVE-SPECIALISED prologue (round 4).  Waves [0, NP) request the activation, run the norm + quantiser among
// themselves (LDS counters, XPre<.., NT>) and only then request their rows; waves [NP, TH/64) request ALL their rows right
// after the first barrier and wait for the image behind them.  In the classic form every wave takes part in the prologue's
// barriers, so at most D1 rows per wave (what the CU's memory queue accepts without blocking) can be in flight while the
// activation arrives and is quantised -- the stream had a hole of 2-2.5 us per launch (profiles/r04_*trace*).
template <int DT, int NJ, int RW, int EPI, int NORM, bool XADD = false, int TH = DEC_THREADS, int NP = 0>
__global__ void __launch_bounds__(TH) k_dec_gemv(const half_t *px, const half_t *pnw, const half_t *pnb, int pcols,
                                                          const uint8_t *pw0, const uint8_t *pw1, int pnblk_grid, int ptotal,
                                                          const DecGemvParams P)
{
    // px / pnw / pnb / pcols repeat P.x / P.norm_w / P.norm_b / P.cols as leading scalar arguments: with
    // -amdgpu-kernarg-preload-count they arrive in SGPRs at wave launch, so the activation requests -- the head of
    // the kernel's critical path -- do not wait for the first scalar load of the argument block.
    // pw0 / pw1 / pnblk / ptotal (round 5) do the same for the WEIGHT requests: pnblk = P.nblk and ptotal = P.total_rows for every
    // launch; pw0 = P.W0[0], pw1 = P.W1 for the single-matrix launches (Wo, W1 / W3, W2), pw0 = null when there are several sets
    // or an expert table (the row addresses then come from the argument block as before).  Every row address of such a launch is then a function of preloaded scalars and
    // the wave's own id: the stream's first request no longer waits ~0.4 us for the argument block's scalar load (r04 trace:
    // "loads issued" at 0.9-1.0 us of the launch; the argument block is cold after every kernel boundary)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(NP == 0 || (NORM != 2 && !XADD && NP * 64 < TH), "wave-specialised prologue: a norm / quantiser prologue, some loader waves");
    // the activation requests go out first, from preloaded arguments only (nothing here waits for a scalar load)
    constexpr int PT = NP ? NP * 64 : TH;                                                 // threads that run the prologue
    constexpr int MAXC = NORM == 2 ? 1 : (NJ * 8 * block_capacity(DT) + PT - 1) / PT;      // cols <= 64 * capacity * NJ: chunks of 8 per thread
    XPre<NORM == 2 ? 0 : NORM, MAXC, XADD, NP * 64> pre;
    if constexpr (NORM != 2) {
        if (NP == 0 || threadIdx.x < PT) pre.issue(px, pnw, pnb, pcols);
        if constexpr (XADD) pre.issue_add(P.x_add, P.x_add_bias, pcols);
    }
    const long long t_start = wall_clock64();
    const XLds L = xlds_carve(smem, pcols);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform (SGPR)
    const int gw = blockIdx.x * (TH / 64) + wave;
    // (the grid size from the preloaded scalar too: gridDim.x is a HIDDEN kernel argument, i.e. one more scalar load of the cold
    //  argument block in front of the first row address)
    const int W = (int)((unsigned)pnblk_grid >> 16) * (TH / 64);
    using Fmt = DecFmt<DT, NJ>;
    constexpr int NM = epi_is_glu(EPI) ? 2 : 1;
    // single-matrix launch: rows, blocks and matrix pointers from the preloaded scalars (see above); the selects below are on a
    // wave-uniform condition the compiler resolves with scalar instructions -- nothing on this path reads the argument block
    const bool single = !epi_is_moe(EPI) && pw0 != nullptr;
    const int nblk_k = pnblk_grid & 0xFFFF, total_rows = ptotal;          // (= P.nblk, P.total_rows for EVERY launch: no select on the argument block)
    const size_t row_bytes = tiled_row_bytes(DT, (size_t)nblk_k);
    const int npass = (total_rows + RW * W - 1) / (RW * W);

    // (P.trace is the first field of the argument block anything here reads: its test sits BEHIND the first weight requests,
    //  trace_on() below)

    // MoE: the expert's matrices, looked up once (uniform scalar loads) from the router's choice
    // (several sets = several router slots in one launch: set i is the expert of slot moe_slot + i)
    const uint8_t *moeW0 = nullptr, *moeW1 = nullptr, *moeW0b = nullptr, *moeW1b = nullptr, *moeW0c = nullptr, *moeW1c = nullptr;
    if constexpr (epi_is_moe(EPI)) {
        const int e = __builtin_amdgcn_readfirstlane(P.moe_sel[P.moe_slot]);
        moeW0 = P.w_table[4 * e + P.moe_tab_off];
        if constexpr (NM == 2) moeW1 = P.w_table[4 * e + P.moe_tab_off + 1];
        moeW0b = moeW0; moeW1b = moeW1; moeW0c = moeW0; moeW1c = moeW1;
        if (P.nsets > 1) {
            const int e1 = __builtin_amdgcn_readfirstlane(P.moe_sel[P.moe_slot + 1]);
            moeW0b = P.w_table[4 * e1 + P.moe_tab_off];
            if constexpr (NM == 2) moeW1b = P.w_table[4 * e1 + P.moe_tab_off + 1];
        }
        if (P.nsets > 2) {
            const int e2 = __builtin_amdgcn_readfirstlane(P.moe_sel[P.moe_slot + 2]);
            moeW0c = P.w_table[4 * e2 + P.moe_tab_off];
            if constexpr (NM == 2) moeW1c = P.w_table[4 * e2 + P.moe_tab_off + 1];
        }
    }
    auto moe_pick = [](int si, const uint8_t *a, const uint8_t *b, const uint8_t *c) { return si == 2 ? c : (si == 1 ? b : a); };
    typename Fmt::W w[NM][RW];
    auto load_rows = [&](int pass, int i0, int i1) {
        // rows past the end are not requested at all: clamped re-reads of the last row used to fill the CU's request
        // window with duplicates (2 of 3 requests of the Wo kernel, 10 % of W1/W3).  `full` is wave-uniform (gw is an
        // SGPR): a wave whose pass is complete -- every wave of most launches -- takes the straight-line path without
        // per-row branches; only the first row of a pass is clamped, so that every wave owns defined registers
        const bool full = (pass * RW + RW - 1) * W + gw < total_rows;
        auto one = [&](int i) {
            const int v = min((pass * RW + i) * W + gw, total_rows - 1);
            if (single) {      // (a uniform branch, not a select: the other arm waits for the argument block)
                w[0][i].load(pw0 + (size_t)v * row_bytes, nblk_k, lane);
                if constexpr (NM == 2) w[1][i].load(pw1 + (size_t)v * row_bytes, nblk_k, lane);
                return;
            }
            const DecRow d = dec_locate(P, v);
            const uint8_t *W0 = epi_is_moe(EPI) ? moe_pick(d.si, moeW0, moeW0b, moeW0c) : d.W0;
            w[0][i].load(W0 + (size_t)d.row * row_bytes, nblk_k, lane);
            if constexpr (NM == 2) { const uint8_t *W1 = epi_is_moe(EPI) ? moe_pick(d.si, moeW1, moeW1b, moeW1c) : d.W1; w[1][i].load(W1 + (size_t)d.row * row_bytes, nblk_k, lane); }
        };
        if (full) {
#pragma unroll
            for (int i = 0; i < RW; i++) { if (i < i0 || i >= i1) continue; one(i); }
        } else {
#pragma unroll
            for (int i = 0; i < RW; i++) {
                if (i < i0 || i >= i1) continue;
                if (i > 0 && (pass * RW + i) * W + gw >= total_rows) continue;
                one(i);
            }
        }
    };
    auto load_pass = [&](int pass) { load_rows(pass, 0, RW); };
    // the residual value(s) lane i adds to row i of a pass are requested right behind that pass's weights (clamped,
    // unconditional): issued from the epilogue they were a second memory round trip at the very end of the kernel
    half_t res = (half_t)0, res2 = (half_t)0;
    auto load_epi = [&](int pass) {
        if constexpr (EPI == EPI_RESIDUAL) {
            const int v = min((pass * RW + min(lane, RW - 1)) * W + gw, total_rows - 1);
            const int row = single ? v : dec_locate(P, v).row;
            res = P.residual[row];
            if (P.residual2) res2 = P.residual2[row];
        }
    };
    // rows requested BEFORE the cooperative prologue: what the CU's memory pipeline accepts
    // without blocking (~32-48 KiB per CU); a barrier behind blocked loads would only
    // release once the slowest wave's requests have been accepted, i.e. late in the stream
    constexpr int D1 = (NM * NJ * Fmt::DW >= 15) ? 1 : 2;

    typename Fmt::X X;
    bool tr = false;
    auto trace_on = [&]() { tr = P.trace != nullptr && threadIdx.x == 64; if (tr) P.trace[blockIdx.x * 8 + 0] = t_start; };
    if constexpr (NORM == 2) {
        // the producing kernel left the quantised activation: a wave's requests are its slice of it, then its rows --
        // no cooperative step, no barrier, no LDS
        const XqImage Q = xq_image_carve(const_cast<half_t *>(px), pcols);      // (px = P.x: the image, not F16 values)
        X.load(Q.codes, Q.scale, Q.xsum, lane, nblk_k);
        load_rows(0, 0, RW);
        load_epi(0);
        trace_on();
        if (tr) { P.trace[blockIdx.x * 8 + 1] = wall_clock64(); P.trace[blockIdx.x * 8 + 2] = wall_clock64(); }
        if (gw >= total_rows) return;
    } else {
        // the CU's memory queue is FIFO across waves: make sure every wave's activation
        // request is queued before ANY wave floods it with weight requests
        if constexpr (NP > 0) {
            if (threadIdx.x == TH - 1) { L.part[130] = 0.0f; L.part[131] = 0.0f; }      // the two LDS counters (bit pattern of +0)
            // the loader waves' first row goes out BEFORE the barrier: its addresses wait for the argument block (~0.4 us, by which
            // time the activation requests of the prologue waves -- issued from preloaded arguments -- are queued), the barrier
            // then costs the prologue waves nothing (they wait ~1 us for the activation anyway) and the stream starts ~0.3 us earlier
            // (NORM == 1 only: the [dim] activation is one or two requests per prologue thread; W2's 11008-value input is six, and weight
            //  requests slipping in between them delayed its quantiser: 8.0 -> 8.7 us)
            constexpr int EARLY = NORM == 1 ? 1 : 0;
            if (EARLY && wave >= NP) load_rows(0, 0, 1);
            trace_on();
            __syncthreads();
            if (wave >= NP) {
                load_rows(0, EARLY, RW);
                load_epi(0);
                if (P.trace != nullptr && threadIdx.x == NP * 64 && NORM == 1) P.trace[blockIdx.x * 8 + 4] = wall_clock64();
            } else {
                if (tr) P.trace[blockIdx.x * 8 + 1] = wall_clock64();
                pre.finish(P.norm_w, P.norm_b, P.multi_base, P.eps, P.cols, L, P.xn_out,
                           (P.trace != nullptr && threadIdx.x == 0) ? P.trace + blockIdx.x * 8 : nullptr);
                load_rows(0, 0, RW);
                load_epi(0);
                if (tr) P.trace[blockIdx.x * 8 + 2] = wall_clock64();
            }
            lds_counter_wait(L.part + 131, NP);
        } else {
        __syncthreads();
        load_rows(0, 0, D1);
        trace_on();
        if (tr) P.trace[blockIdx.x * 8 + 1] = wall_clock64();
        if constexpr (XADD) pre.apply_add(P.x_add_bias, P.cols, P.xsum_out);
        pre.finish(P.norm_w, P.norm_b, P.multi_base, P.eps, P.cols, L, P.xn_out,
                   (P.trace != nullptr && threadIdx.x == 0) ? P.trace + blockIdx.x * 8 : nullptr);
        load_rows(0, D1, RW);
        load_epi(0);
        if (tr) P.trace[blockIdx.x * 8 + 2] = wall_clock64();
        }
        if (gw >= total_rows) return;
        X.load(L.codes, L.scale, L.xsum, lane, nblk_k);
    }
    if (tr) P.trace[blockIdx.x * 8 + 3] = wall_clock64();

    for (int pass = 0; pass < npass; pass++) {
        if (pass > 0) { load_pass(pass); load_epi(pass); }
        float a[NM][RW];
#pragma unroll
        for (int i = 0; i < RW; i++)
#pragma unroll
            for (int m = 0; m < NM; m++) a[m][i] = w[m][i].dot(X);
#pragma unroll
        for (int i = 0; i < RW; i++)
#pragma unroll
            for (int m = 0; m < NM; m++) a[m][i] = wave_sum(a[m][i]);
        // lane i finishes row i
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int i = 0; i < RW; i++) {
            if (lane == i) { a0 = a[0][i]; if constexpr (NM == 2) a1 = a[1][i]; }
        }
        const int v = (pass * RW + lane) * W + gw;
        if (lane < RW && v < total_rows) {
            dec_finish_row<EPI>(P, dec_locate(P, v), a0, a1, res, res2);
        }
    }
    if (tr) P.trace[blockIdx.x * 8 + 7] = wall_clock64();
}

// Rows longer than a lane's register image (cols > 64 * NJmax blocks: w2 of 34B-70B models, 20480-32768 columns):
// the row is walked in chunks of 64*NJ blocks; the (pass, chunk) sequence is software-pipelined over two register
// sets, the activation slice of a chunk is re-read from LDS, and every lane continues ONE accumulation chain over
// its blocks in ascending order (dot(X, acc)), so results are bit-identical to the single-chunk kernel / op path.
// No norm prologue and no GLU pair (only Wo / W2-type matrices get this long).
template <int DT, int NJ, int RW, int EPI>
__global__ void __launch_bounds__(DEC_THREADS) k_dec_gemv_long(const DecGemvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const XLds L = xlds_carve(smem, P.cols);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * DEC_WAVES + wave;
    const int W = gridDim.x * DEC_WAVES;
    using Fmt = DecFmt<DT, NJ>;
    const size_t row_bytes = tiled_row_bytes(DT, (size_t)P.nblk);
    const int npass = (P.total_rows + RW * W - 1) / (RW * W);
    const int nchunk = (P.nblk + 64 * NJ - 1) / (64 * NJ);
    const int Q = npass * nchunk;
    const uint8_t *moeW0 = nullptr;
    if constexpr (epi_is_moe(EPI)) moeW0 = P.w_table[4 * __builtin_amdgcn_readfirstlane(P.moe_sel[P.moe_slot]) + P.moe_tab_off];
    typename Fmt::W wa[RW], wb[RW];
    auto load_q = [&](typename Fmt::W (&w)[RW], int q, int i0, int i1) {
        const int pass = q / nchunk, chunk = q - pass * nchunk;
#pragma unroll
        for (int i = 0; i < RW; i++) {
            if (i < i0 || i >= i1) continue;
            const int v = min((pass * RW + i) * W + gw, P.total_rows - 1);
            const DecRow d = dec_locate(P, v);
            const uint8_t *W0 = epi_is_moe(EPI) ? moeW0 : d.W0;
            w[i].load(W0 + (size_t)d.row * row_bytes, P.nblk, lane, chunk * 64 * NJ);
        }
    };
    {
        XPre<0, 8> pre;                       // up to 32768 columns
        pre.issue(P.x, nullptr, nullptr, P.cols);
        __syncthreads();
        load_q(wa, 0, 0, 1);
        pre.finish(nullptr, nullptr, 0.0f, P.eps, P.cols, L, nullptr, nullptr);
        load_q(wa, 0, 1, RW);
    }
    if (gw >= P.total_rows) return;
    float a[RW];
#pragma unroll
    for (int i = 0; i < RW; i++) a[i] = 0.0f;
    auto compute = [&](typename Fmt::W (&w)[RW], int q) {
        const int pass = q / nchunk, chunk = q - pass * nchunk;
        typename Fmt::X X;
        X.load(L.codes, L.scale, L.xsum, lane, P.nblk, chunk * 64 * NJ);
#pragma unroll
        for (int i = 0; i < RW; i++) a[i] = w[i].dot(X, a[i]);
        if (chunk + 1 < nchunk) return;
#pragma unroll
        for (int i = 0; i < RW; i++) a[i] = wave_sum(a[i]);
        float a0 = 0.0f;
#pragma unroll
        for (int i = 0; i < RW; i++) { if (lane == i) a0 = a[i]; a[i] = 0.0f; }
        const int v = (pass * RW + lane) * W + gw;
        if (lane < RW && v < P.total_rows) {
            const DecRow d = dec_locate(P, v);
            half_t res = (half_t)0, res2 = (half_t)0;
            if constexpr (EPI == EPI_RESIDUAL) { res = P.residual[d.row]; if (P.residual2) res2 = P.residual2[d.row]; }
            dec_finish_row<EPI>(P, d, a0, 0.0f, res, res2);
        }
    };
    for (int q = 0; q < Q; q += 2) {
        if (q + 1 < Q) load_q(wb, q + 1, 0, RW);
        compute(wa, q);
        if (q + 1 < Q) {
            if (q + 2 < Q) load_q(wa, q + 2, 0, RW);
            compute(wb, q + 1);
        }
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
    const int gw = blockIdx.x * DEC_WAVES + __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = gridDim.x * DEC_WAVES;
    const int nbatch = (P.rows + R - 1) / R;
    const int chunks = P.cols >> 3;

    auto load_batch = [&](u32x4 (&dst)[R][NJ], int b) {
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const int row = min(b * R + rr, P.rows - 1);          // clamped, unconditional (see WRowQ4::load)
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int c = min(lane + 64 * j, chunks - 1);
                dst[rr][j] = nt_load<u32x4>(reinterpret_cast<const u32x4 *>(P.W + (size_t)row * P.cols) + c);
            }
        }
    };
    u32x4 cur[R][NJ];
    load_batch(cur, gw);

    for (int c = tid; c < chunks; c += DEC_THREADS)
        *reinterpret_cast<half8_t *>(xn + (size_t)c * 8) = *reinterpret_cast<const half8_t *>(P.x + (size_t)c * 8);
    float scale = 1.0f;
    if constexpr (NORM == 1) {
        // canonical sum of squares (ifa_math.h): a wave butterfly per 64 chunks, groups added in ascending order
        const int ngroups = (chunks + 63) >> 6;
        __syncthreads();
        for (int g = tid >> 6; g < ngroups; g += DEC_WAVES) {
            const int c = 64 * g + lane;
            half8_t v8;
#pragma unroll
            for (int i = 0; i < 8; i++) v8[i] = (half_t)0;
            if (c < chunks) v8 = *reinterpret_cast<const half8_t *>(xn + (size_t)c * 8);
            const float pg = wave_sum(rms_chunk_sq(v8));
            if (lane == 0) part[g] = pg;
        }
        __syncthreads();
        scale = rms_scale_of(rms_total(part, ngroups), P.cols, P.eps);
        for (int c = tid; c < chunks; c += DEC_THREADS) {
            half8_t xv = *reinterpret_cast<const half8_t *>(xn + (size_t)c * 8);
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
            *reinterpret_cast<half8_t *>(xn + (size_t)c * 8) = xv;
            if (P.xn_out && blockIdx.x == 0) *reinterpret_cast<half8_t *>(P.xn_out + (size_t)c * 8) = xv;
        }
    } else {
        if (P.xn_out && blockIdx.x == 0)
            for (int c = tid; c < chunks; c += DEC_THREADS)
                *reinterpret_cast<half8_t *>(P.xn_out + (size_t)c * 8) = *reinterpret_cast<const half8_t *>(xn + (size_t)c * 8);
    }
    __syncthreads();
    if (gw >= nbatch) return;
    u32x4 xr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int c = lane + 64 * j;
        xr[j] = u32x4{0, 0, 0, 0};
        if (c < chunks) xr[j] = *reinterpret_cast<const u32x4 *>(xn + (size_t)c * 8);
    }
    for (int b = gw; b < nbatch; b += W) {
        u32x4 nxt[R][NJ];
        load_batch(nxt, b + W);
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; j++) acc = dot8_f16(cur[rr][j], xr[j], acc);
            acc = wave_sum(acc);
            if (lane == 0 && b * R + rr < P.rows) P.logits[b * R + rr] = f2h(acc);
        }
#pragma unroll
        for (int rr = 0; rr < R; rr++)
#pragma unroll
            for (int j = 0; j < NJ; j++) cur[rr][j] = nxt[rr][j];
    }
}

// (the decode attention kernels: ifa_decode_attn.h)

// ------------------------------------------------------------- small kernels
// state[0] = current token id, state[1] = its position, state[2] = steps done; state[3..6] = excluded ids (see
// k_dec_argmax_advance);
// state[8 + i] = i-th generated token of the current launch batch.
// Also fills the step's RoPE table: tab[c] = (cos, sin) of pos * theta_scale^c,
// the same expression rope_rotate() evaluates per element (ifa_math.h).
// embd_scale != 0: TensorOpr::LinearNorm on the decoder input (has_embedding_linear_norm, inference_worker.cc:447-451; tensor_opr.cu:482-497
// = Tensor_Scale_Kernel: half((float)e * scale)), fused into the row copy
__device__ __forceinline__ u32x4 embd_row_scale(u32x4 v, float embd_scale)
{
    if (embd_scale == 0.0f) return v;
    half8_t h = __builtin_bit_cast(half8_t, v);
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = f2h(h2f(h[i]) * embd_scale);
    return __builtin_bit_cast(u32x4, h);
}

// batched step: embedding row of every query's token (grid (x, queries)) and its RoPE table rope_tab[query][head_dim]
static __global__ void __launch_bounds__(256) k_dec_batch_gather(const half_t *__restrict__ embd, const int *__restrict__ tokens,
                                                                 const int *__restrict__ positions, int dim, int vocab, half_t *__restrict__ x,
                                                                 float *__restrict__ rope_tab, int head_dim, float theta, int rope_dims,
                                                                 float embd_scale)
{
    const int b = blockIdx.y;
    const int tok = min(max(tokens[b], 0), vocab - 1);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < dim / 8; c += gridDim.x * blockDim.x)
        reinterpret_cast<u32x4 *>(x + (size_t)b * dim)[c] = embd_row_scale(reinterpret_cast<const u32x4 *>(embd + (size_t)tok * dim)[c], embd_scale);
    if (blockIdx.x == 0 && rope_tab) {
        const int pos = positions[b];
        for (int c = threadIdx.x; c < head_dim / 2; c += blockDim.x) {
            float cs, sn;
            rope_angle(c, pos, theta, rope_dims, cs, sn);
            rope_tab[(size_t)b * head_dim + 2 * c] = cs; rope_tab[(size_t)b * head_dim + 2 * c + 1] = sn;
        }
    }
}

static __global__ void __launch_bounds__(256) k_dec_gather(const half_t *__restrict__ embd, const int *__restrict__ state,
                                                    int dim, int vocab, half_t *__restrict__ x,
                                                    float *__restrict__ rope_tab, int head_dim, float theta,
                                                    int rope_dims, float embd_scale)
{
    if (embd) {      // null: the layer input arrives from the previous pipeline stage, only the RoPE table is needed
        int tok = state[0];
        tok = min(max(tok, 0), vocab - 1);
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < dim / 8; c += gridDim.x * blockDim.x)
            reinterpret_cast<u32x4 *>(x)[c] = embd_row_scale(reinterpret_cast<const u32x4 *>(embd + (size_t)tok * dim)[c], embd_scale);
    }
    if (blockIdx.x == 0 && rope_tab) {
        const int pos = state[1];
        for (int c = threadIdx.x; c < head_dim / 2; c += blockDim.x) {
            float cs, sn;
            rope_angle(c, pos, theta, rope_dims, cs, sn);
            rope_tab[2 * c] = cs; rope_tab[2 * c + 1] = sn;
        }
    }
}

// greedy top-1 over the logits (first maximum wins); writes the token ring and advances the state.
// state[3] = number of excluded ids (<= 3), state[4..6] = the ids GetSortedTopK never offers to the queue (the
// vocabulary's unk id, Invalid-type tokens: sampling_strategy.cc:281-297)
static __global__ void __launch_bounds__(1024) k_dec_argmax_advance(const half_t *__restrict__ v, int n, int *__restrict__ state,
                                                             int ring)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    // the state words through the constant address space: ONE scalar request next to the scan's vector loads (as vector loads they
    // were two dependent cold round trips in front of the scan and a third behind it)
    const __attribute__((address_space(4))) int *cs = (const __attribute__((address_space(4))) int *)state;
    const int st_pos = cs[1], st_step = cs[2];
    const int ne = min(max(cs[3], 0), 3);
    const int x4 = cs[4], x5 = cs[5], x6 = cs[6];
    const int e0 = ne > 0 ? x4 : -1, e1 = ne > 1 ? x5 : -1, e2 = ne > 2 ? x6 : -1;
    float best = -INFINITY; int besti = 0x7FFFFFFF;
    argmax_scan(v, (size_t)n, e0, e1, e2, (int)threadIdx.x, (int)blockDim.x, best, besti);
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
        const int step = st_step;
        state[8 + (step % ring)] = besti;
        state[0] = besti;
        state[1] = st_pos + 1;
        state[2] = step + 1;
    }
}

} // namespace ifa
