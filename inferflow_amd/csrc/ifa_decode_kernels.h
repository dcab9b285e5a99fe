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
__device__ __forceinline__ void dec_finish_row(const DecGemvParams &P, const DecRow d, float a0, float a1, half_t res = (half_t)0,
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
    d.y[row] = y;
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
                                                          const DecGemvParams P)
{
    // px / pnw / pnb / pcols repeat P.x / P.norm_w / P.norm_b / P.cols as leading scalar arguments: with
    // -amdgpu-kernarg-preload-count they arrive in SGPRs at wave launch, so the activation requests -- the head of
    // the kernel's critical path -- do not wait for the first scalar load of the argument block
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
    const int W = gridDim.x * (TH / 64);
    using Fmt = DecFmt<DT, NJ>;
    const size_t row_bytes = tiled_row_bytes(DT, (size_t)P.nblk);
    constexpr int NM = epi_is_glu(EPI) ? 2 : 1;
    const int npass = (P.total_rows + RW * W - 1) / (RW * W);

    const bool tr = P.trace != nullptr && threadIdx.x == 64;
    if (tr) P.trace[blockIdx.x * 8 + 0] = t_start;

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
        const bool full = (pass * RW + RW - 1) * W + gw < P.total_rows;
        auto one = [&](int i) {
            const int v = min((pass * RW + i) * W + gw, P.total_rows - 1);
            const DecRow d = dec_locate(P, v);
            const uint8_t *W0 = epi_is_moe(EPI) ? moe_pick(d.si, moeW0, moeW0b, moeW0c) : d.W0;
            w[0][i].load(W0 + (size_t)d.row * row_bytes, P.nblk, lane);
            if constexpr (NM == 2) { const uint8_t *W1 = epi_is_moe(EPI) ? moe_pick(d.si, moeW1, moeW1b, moeW1c) : d.W1; w[1][i].load(W1 + (size_t)d.row * row_bytes, P.nblk, lane); }
        };
        if (full) {
#pragma unroll
            for (int i = 0; i < RW; i++) { if (i < i0 || i >= i1) continue; one(i); }
        } else {
#pragma unroll
            for (int i = 0; i < RW; i++) {
                if (i < i0 || i >= i1) continue;
                if (i > 0 && (pass * RW + i) * W + gw >= P.total_rows) continue;
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
            const int v = min((pass * RW + min(lane, RW - 1)) * W + gw, P.total_rows - 1);
            const int row = dec_locate(P, v).row;
            res = P.residual[row];
            if (P.residual2) res2 = P.residual2[row];
        }
    };
    // rows requested BEFORE the cooperative prologue: what the CU's memory pipeline accepts
    // without blocking (~32-48 KiB per CU); a barrier behind blocked loads would only
    // release once the slowest wave's requests have been accepted, i.e. late in the stream
    constexpr int D1 = (NM * NJ * Fmt::DW >= 15) ? 1 : 2;

    typename Fmt::X X;
    if constexpr (NORM == 2) {
        // the producing kernel left the quantised activation: a wave's requests are its slice of it, then its rows --
        // no cooperative step, no barrier, no LDS
        const XqImage Q = xq_image_carve(const_cast<half_t *>(px), pcols);      // (px = P.x: the image, not F16 values)
        X.load(Q.codes, Q.scale, Q.xsum, lane, P.nblk);
        load_rows(0, 0, RW);
        load_epi(0);
        if (tr) { P.trace[blockIdx.x * 8 + 1] = wall_clock64(); P.trace[blockIdx.x * 8 + 2] = wall_clock64(); }
        if (gw >= P.total_rows) return;
    } else {
        // the CU's memory queue is FIFO across waves: make sure every wave's activation
        // request is queued before ANY wave floods it with weight requests
        if constexpr (NP > 0) {
            if (threadIdx.x == TH - 1) { L.part[130] = 0.0f; L.part[131] = 0.0f; }      // the two LDS counters (bit pattern of +0)
            __syncthreads();
            if (wave >= NP) {
                load_rows(0, 0, RW);
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
        if (tr) P.trace[blockIdx.x * 8 + 1] = wall_clock64();
        if constexpr (XADD) pre.apply_add(P.x_add_bias, P.cols, P.xsum_out);
        pre.finish(P.norm_w, P.norm_b, P.multi_base, P.eps, P.cols, L, P.xn_out,
                   (P.trace != nullptr && threadIdx.x == 0) ? P.trace + blockIdx.x * 8 : nullptr);
        load_rows(0, D1, RW);
        load_epi(0);
        if (tr) P.trace[blockIdx.x * 8 + 2] = wall_clock64();
        }
        if (gw >= P.total_rows) return;
        X.load(L.codes, L.scale, L.xsum, lane, P.nblk);
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
        if (lane < RW && v < P.total_rows) {
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

// ------------------------------------------------------------------ attention
struct DecAttnParams {
    const half_t *q;           // [heads*head_dim]   pre-RoPE
    const half_t *k_new;       // [kv_heads*head_dim] pre-RoPE
    const half_t *v_new;       // [kv_heads*head_dim]
    uint8_t *kcache, *vcache;  // [max_ctx][kv_row_bytes]
    const int *state;          // state[1] = position of the new token
    const float *rope_tab;     // [head_dim/2][2] cos,sin of this step (k_dec_gather)
    int heads, kv_heads, kv_q8;
    float kq_scale;
    int rope_order, rope_cols;
    int alibi, alibi_base, alibi_total;
    half_t *out;               // [heads*head_dim]
    int max_ctx;
    int8_t *xq;                // optional XqImage of `out` (Q8_B32T2, the quantiser the Wo GEMV would run in its prologue)
    long long *trace;          // optional [heads][8] wall-clock stamps (100 MHz) for tuning
    // batched step (k_dec_attn<.., BATCH = true>, grid (heads, queries)): query b reads q|k|v at q + b * q_stride, its cache
    // pointers and context from batch_rows[b], its RoPE pairs at rope_tab + b * head_dim, and writes out + b * heads * head_dim
    const void *batch_rows;    // DecAttnBatchRow[queries]
    int q_stride;
};
struct DecAttnBatchRow { const uint8_t *kc, *vc; int n_ctx, pad; };

// Quantize(kqv_merged) (inference_worker.cc:1339-1346) done where the vector is produced: a head is HD/32 whole
// Q8_B32T2 blocks, so the blocks are local to the head's workgroup and the codes are those of the Alg2 quantizer
// (tensor_quant.h:44-82) bit for bit.  Called by threads [0, HD) of the workgroup of head h with their output value.
template <int HD>
__device__ __forceinline__ void dec_attn_emit_q8(int8_t *xq, int cols, int h, int d, half_t yh)
{
    const XqImage Q = xq_image_carve(xq, cols);
    const float val = h2f(yh);
    const float mx = half_wave_max(fabsf(val));
    const float qs = mx / 127;
    const int qv = q8_round_div1(val, qs);
    const int sum = half_wave_sum_i32(qv);
    Q.codes[(size_t)h * HD + d] = (int8_t)qv;
    if ((d & 31) == 0) {
        const int blk = (h * HD + d) >> 5;
        Q.scale[blk] = h2f(f2h(qs));
        Q.xsum[blk] = (float)sum;
    }
}

// rotate one pair with a precomputed (cos, sin); same expressions as rope_rotate
__device__ __forceinline__ void rope_apply(half_t *row, int col, float c, float s, int order, int rope_cols)
{
    int i0, i1;
    if (order == 2) { if (2 * col >= rope_cols) return; i0 = col; i1 = col + rope_cols / 2; }
    else { i0 = 2 * col; i1 = 2 * col + 1; }
    const float x0 = h2f(row[i0]), x1 = h2f(row[i1]);
    float a = x0 * c, bq = x1 * s, d = x0 * s, e = x1 * c;
    row[i0] = f2h(a - bq);
    row[i1] = f2h(d + e);
}

// One workgroup (256 threads) per query head.
//  * K rows: one key per lane, whole row slice in registers (loads issued at
//    kernel entry, before q/k/v staging), fp32 fma in d order == Gemm_Alg2 order,
//    so S is bit-exact with the reference arithmetic.
//  * V rows: thread (d-group of 8, key residue mod 256/(HD/8)); loads for the
//    first key chunk are also issued at entry.  Partials are combined in a fixed
//    order through LDS.
//  * the new token's K/V never round-trip through HBM: they come from LDS and are
//    written to the cache by the first head of each KV group.
// pq / pkc / pvc / pheads / pkvh repeat the new token's q|k|v vector (ONE buffer: q, then k at + heads*HD, then v at
// + (heads + kv_heads)*HD), the layer's K / V cache and the head counts as leading scalar arguments: a by-value struct is
// fetched with scalar loads (cold after every kernel boundary), these 8 dwords are preloaded into SGPRs at wave launch, so
// the q / k / v values and the first 256 K / V rows are requested with the kernel's first instructions.  The caches hold
// at least DEC_ATTN_MIN_ROWS rows (the engine pads the allocation), so that first chunk needs no clamp.
constexpr int DEC_ATTN_MIN_ROWS = 256;

template <int HD, bool Q8, bool BATCH = false>
__global__ void __launch_bounds__(256) k_dec_attn(const half_t *pq, const uint8_t *pkc, const uint8_t *pvc, int pheads, int pkvh,
                                                  const DecAttnParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int pos_b = 0;
    if constexpr (BATCH) {      // the query's cache pointers and position come from the step's table (one scalar fetch)
        const DecAttnBatchRow br = reinterpret_cast<const DecAttnBatchRow *>(P.batch_rows)[blockIdx.y];
        pkc = br.kc; pvc = br.vc; pos_b = br.n_ctx - 1;
        pq += (size_t)blockIdx.y * P.q_stride;
    }
    // The cache rows are read through pointers whose address space is STATED (IFA_GP): in the batched step they come from the
    // table in memory, and the FLAT loads the compiler emits for a pointer of unknown origin also count on lgkmcnt -- every LDS
    // wait of the score / softmax phases then waited for the K / V requests in flight
#define IFA_GP(T, p) ((const __attribute__((address_space(1))) T *)(p))
    uint8_t *const kcw = BATCH ? const_cast<uint8_t *>(pkc) : P.kcache;      // the cache rows this workgroup may write
    uint8_t *const vcw = BATCH ? const_cast<uint8_t *>(pvc) : P.vcache;
    const float *const rope_tab = BATCH ? P.rope_tab + (size_t)blockIdx.y * HD : P.rope_tab;
    half_t *const outp = BATCH ? P.out + (size_t)blockIdx.y * pheads * HD : P.out;
    static_assert(HD % 8 == 0 && HD <= 128 && (!Q8 || HD % 32 == 0), "head size: multiples of 8 up to 128 (Q8 rows: whole 32-blocks)");
    constexpr int DG = HD / 8;            // threads covering one V row (8 dims each)
    constexpr int NSPLIT = 256 / DG;      // key residues handled in parallel (head sizes 48 / 80 / 96: the last 256 % DG threads idle)
    half_t *qs = reinterpret_cast<half_t *>(smem);                 // [HD] rotated q
    half_t *kn = qs + HD;                                          // [HD] rotated (and Q8 round-tripped) new k
    half_t *vn = kn + HD;                                          // [HD] new v (Q8 round-tripped)
    float *red = reinterpret_cast<float *>(vn + HD);               // [16]
    float *opart = red + 16;                                       // [NSPLIT][HD]
    half_t *S = reinterpret_cast<half_t *>(opart + NSPLIT * HD);   // [n_ctx]
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = pheads / pkvh;
    const int kvh = h / group;
    const bool writer = (h % group) == 0;
    const int kv_dim = pkvh * HD;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const size_t head_off = Q8 ? (size_t)((kvh * HD) / 32) * 34 : (size_t)kvh * HD * 2;

    // ---- issue the new token's values, then the first chunk of K (key = tid) and V loads before anything else.  Keys
    // past the context are loaded too (never used) instead of being masked off: loads under an exec mask make every later
    // wait a vmcnt(0), and the position comes from device memory -- nothing here waits for it, or for the argument block
    // Q8 rows: a head's slice is (HD/32)*34 bytes, 8-byte aligned for HD=128, 4-byte for HD=64, 2-byte for HD=32
    constexpr int KBYTES = (HD / 32) * 34;
    constexpr int KALIGN = HD == 128 ? 8 : (HD == 64 ? 4 : 2);
    uint32_t kreg[Q8 ? 1 : HD / 2];
    uint32_t kq32[(Q8 && KALIGN >= 4) ? KBYTES / 4 : 1];
    uint16_t kq16[(Q8 && KALIGN < 4) ? KBYTES / 2 : 1];
    auto load_k = [&](int j) {
        const uint8_t *rowp = pkc + (size_t)j * row_bytes + head_off;
        if constexpr (Q8) {
            if constexpr (KALIGN == 8) {
#pragma unroll
                for (int i = 0; i < KBYTES / 8; i++) {
                    const u32x2 t = IFA_GP(u32x2, rowp)[i];
                    kq32[2 * i] = t[0]; kq32[2 * i + 1] = t[1];
                }
            } else if constexpr (KALIGN == 4) {
#pragma unroll
                for (int i = 0; i < KBYTES / 4; i++) kq32[i] = IFA_GP(uint32_t, rowp)[i];
            } else {
#pragma unroll
                for (int i = 0; i < KBYTES / 2; i++) kq16[i] = IFA_GP(uint16_t, rowp)[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < HD / 8; i++) {
                const u32x4 t = IFA_GP(u32x4, rowp)[i];
                kreg[4 * i] = t[0]; kreg[4 * i + 1] = t[1]; kreg[4 * i + 2] = t[2]; kreg[4 * i + 3] = t[3];
            }
        }
    };
    // byte B (compile-time) of the Q8 slice held in registers
    auto kbyte = [&](int B) -> uint32_t {
        if constexpr (KALIGN >= 4) return (kq32[B >> 2] >> (8 * (B & 3))) & 0xFFu;
        else return (kq16[B >> 1] >> (8 * (B & 1))) & 0xFFu;
    };
    // the new token's q / k / v values and this step's (cos, sin) pair FIRST: loads return in issue order, and behind the
    // 128 KB of K / V rows below these few bytes arrived 1.5 us later than they had to (phase stamps)
    const int dq = min(tid, HD - 1);
    const half_t q_in = pq[(size_t)h * HD + dq], k_in = pq[(size_t)(pheads + kvh) * HD + dq], v_in = pq[(size_t)(pheads + pkvh + kvh) * HD + dq];
    // (tid < DEC_ATTN_MIN_ROWS <= rows of the cache.)  Batched step: the position arrived with the cache pointers, so rows past
    // the context are clamped to the last one (duplicate addresses: one cache line) -- unclamped, every (head, query) workgroup
    // pulled 2 x 64 KB of cache rows whatever its context: 134 MB per layer at 32 queries, the whole cost of that launch
    load_k(BATCH ? min(tid, pos_b) : tid);
    // this step's (cos, sin) pair of the thread that will rotate: requested BETWEEN the K and the V rows -- behind both it was the
    // newest request, and the rotation waited (vmcnt(0)) for the whole prefetch; unconditional (a valid dummy address without RoPE)
    float rope_cs = 1.0f, rope_sn = 0.0f;
    {
        const int c = min(tid < HD / 2 ? tid : tid - HD / 2, HD / 2 - 1);
        const float *rt = rope_tab ? rope_tab : reinterpret_cast<const float *>(pq);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 cs = *IFA_GP(f32x2, rt + 2 * c);
        if (P.rope_order != 0) { rope_cs = cs[0]; rope_sn = cs[1]; }
    }
    const int dg = tid % DG, sp = tid / DG;
    const bool vact = (256 % DG == 0) || sp < NSPLIT;   // this thread takes part in P.V
    constexpr int VPRE = 256 / NSPLIT;                  // prefetched V keys per thread: j = sp + NSPLIT*i (256 keys)
    u32x4 vreg[Q8 ? 1 : VPRE];
    uint16_t vq[Q8 ? VPRE : 1][5];                      // Q8: {scale, 4 x 2 codes} of this thread's 8 dims, 2-byte aligned
    const size_t vq_off = head_off + (size_t)(dg / 4) * 34;
#pragma unroll
    for (int i = 0; i < VPRE; i++) {
        const int j = min(sp + NSPLIT * i, BATCH ? pos_b : DEC_ATTN_MIN_ROWS - 1);
        if constexpr (!Q8) {
            vreg[i] = IFA_GP(u32x4, pvc + (size_t)j * row_bytes + head_off)[dg];
        } else {
            // the thread's 8 codes as ONE 8-byte request at a 2-byte-aligned address (global memory takes unaligned dwords): five
            // 2-byte requests per key were 80 load instructions per thread ahead of everything else in the kernel
            const auto *blk = IFA_GP(uint16_t, pvc + (size_t)j * row_bytes + vq_off);
            typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
            vq[i][0] = blk[0];
            const u32x2_a2 cw = *IFA_GP(u32x2_a2, blk + 1 + (dg % 4) * 4);
            vq[i][1] = (uint16_t)(cw[0] & 0xFFFFu); vq[i][2] = (uint16_t)(cw[0] >> 16);
            vq[i][3] = (uint16_t)(cw[1] & 0xFFFFu); vq[i][4] = (uint16_t)(cw[1] >> 16);
        }
    }

    // ---- everything below may wait for the argument block: the position, this step's (cos, sin) pair of the thread that
    // will rotate (requested now, used after the staging barrier)
    const bool tr = P.trace != nullptr && tid == 0;
    if (tr) P.trace[h * 8 + 0] = wall_clock64();
    // (a SCALAR load through the constant address space: as a vector load it was the newest request of the wave, and waiting for it
    //  -- vmcnt(0) -- meant waiting for every K / V row requested above before the new token's values could even be staged)
    const int pos = BATCH ? pos_b : *(const __attribute__((address_space(4))) int *)(P.state + 1);
    const int n_ctx = pos + 1;
    // ---- stage q, k_new, v_new; RoPE on q and k (TensorOpr::PositionEmbedding, F16 in/out)
    if (tid < HD) { qs[tid] = q_in; kn[tid] = k_in; vn[tid] = v_in; }
    __syncthreads();
    if (tr) P.trace[h * 8 + 1] = wall_clock64();
    if (P.rope_order != 0) {
        if (tid < HD) {     // threads [0,HD/2) rotate q pairs, [HD/2,HD) rotate k pairs
            const int c = tid < HD / 2 ? tid : tid - HD / 2;
            rope_apply(tid < HD / 2 ? qs : kn, c, rope_cs, rope_sn, P.rope_order, P.rope_cols);
        }
        __syncthreads();
    }
    // ---- KV store of the new row (LayerKVCache::SetKRows/SetVRows, kv_cache.cc:159-249)
    if constexpr (Q8) {
        constexpr int NB = HD / 32;
        // one 32-value block per HALF wave (HD = 128: the 4 + 4 blocks of the new k and v rows on the 8 half waves at once; the
        // block maximum by DPP + one permute): two blocks per wave one after the other on 32 lanes with five LDS permutes each was
        // 0.9 us of this kernel.  Same operations per element as before (the maximum does not depend on the order).
        for (int b = wave * 2 + (lane >> 5); b < 2 * NB; b += 8) {
            half_t *src = b < NB ? kn : vn;
            const int bb = b < NB ? b : b - NB;
            const int l32 = lane & 31;
            {
                const float val = h2f(src[bb * 32 + l32]);
                const float mx = half_wave_max(fabsf(val));
                const float sc = mx / 127;
                int qv = sc <= 0.000001f ? 0 : (int)roundf(val / sc);
                qv = min(max(qv, -128), 127);
                const half_t sch = f2h(sc);
                if (writer) {
                    uint8_t *cache = b < NB ? kcw : vcw;
                    uint8_t *blk = cache + (size_t)pos * row_bytes + head_off + (size_t)bb * 34;
                    blk[2 + l32] = (uint8_t)(int8_t)qv;
                    if (l32 == 0) *reinterpret_cast<uint16_t *>(blk) = __builtin_bit_cast(uint16_t, sch);
                }
                src[bb * 32 + l32] = f2h((float)qv * h2f(sch));   // dequantised value, as GetKRows returns it
            }
        }
        __syncthreads();
    } else {
        if (writer && tid < HD) {
            reinterpret_cast<half_t *>(kcw + (size_t)pos * row_bytes + head_off)[tid] = kn[tid];
            reinterpret_cast<half_t *>(vcw + (size_t)pos * row_bytes + head_off)[tid] = vn[tid];
        }
    }

    // ---- scores: one key per lane, fp32 fma in d order (Gemm_Alg2_Kernel order, products exact)
    if (tr) P.trace[h * 8 + 2] = wall_clock64();
    const float alpha = 1.0f / sqrtf((float)HD) / P.kq_scale;
    const float mk = P.alibi ? alibi_slope(h + P.alibi_base, P.alibi_total) : 0.0f;
    float lmax = -INFINITY;
    for (int j = tid; j < n_ctx; j += 256) {
        float c = 0.0f;
        if (Q8 && j == pos) {
            // the new token's (round-tripped) key: wide LDS reads into registers, then the same chain -- a scalar loop
            // over LDS kept the whole workgroup waiting at the next barrier for ~0.7 us
            uint32_t knr[HD / 2];
#pragma unroll
            for (int i = 0; i < HD / 8; i++) {
                const u32x4 t = reinterpret_cast<const u32x4 *>(kn)[i];
                knr[4 * i] = t[0]; knr[4 * i + 1] = t[1]; knr[4 * i + 2] = t[2]; knr[4 * i + 3] = t[3];
            }
#pragma unroll
            for (int i = 0; i < HD / 2; i++) {
                const half2_t k2 = __builtin_bit_cast(half2_t, knr[i]);
                c = __builtin_fmaf(h2f(qs[2 * i]), (float)k2[0], c);
                c = __builtin_fmaf(h2f(qs[2 * i + 1]), (float)k2[1], c);
            }
        } else {
            if constexpr (!Q8) {
                // the new token's key comes from LDS into the same registers (wide reads) and takes the common path: a
                // scalar loop over LDS here kept the whole workgroup waiting at the next barrier for ~0.7 us
                if (j == pos) {
#pragma unroll
                    for (int i = 0; i < HD / 8; i++) {
                        const u32x4 t = reinterpret_cast<const u32x4 *>(kn)[i];
                        kreg[4 * i] = t[0]; kreg[4 * i + 1] = t[1]; kreg[4 * i + 2] = t[2]; kreg[4 * i + 3] = t[3];
                    }
                } else if (j >= 256) load_k(j);      // later chunks: load now (first chunk was prefetched)
            } else if (j >= 256) load_k(j);
            if constexpr (Q8) {
#pragma unroll
                for (int b = 0; b < HD / 32; b++) {
                    const uint16_t scb = (uint16_t)(kbyte(b * 34) | (kbyte(b * 34 + 1) << 8));
                    if constexpr (KALIGN >= 4) {
                        // four codes per dword (q8x4_dequant_h); a block's codes start at byte 34 b + 2 of the slice
                        const half_t sch = __builtin_bit_cast(half_t, scb);
                        const half2_t sc2 = {sch, sch};
#pragma unroll
                        for (int w4 = 0; w4 < 8; w4++) {
                            const int B0 = b * 34 + 2 + 4 * w4;
                            const uint32_t cw = (B0 & 3) == 0 ? kq32[B0 >> 2]
                                : __builtin_amdgcn_alignbyte(kq32[(B0 >> 2) + 1 < (int)(sizeof(kq32) / 4) ? (B0 >> 2) + 1 : (B0 >> 2)], kq32[B0 >> 2], (B0 & 3));
                            half2_t lo, hi;
                            q8x4_dequant_h(cw, sc2, lo, hi);
                            c = __builtin_fmaf(h2f(qs[b * 32 + 4 * w4]), (float)lo[0], c);
                            c = __builtin_fmaf(h2f(qs[b * 32 + 4 * w4 + 1]), (float)lo[1], c);
                            c = __builtin_fmaf(h2f(qs[b * 32 + 4 * w4 + 2]), (float)hi[0], c);
                            c = __builtin_fmaf(h2f(qs[b * 32 + 4 * w4 + 3]), (float)hi[1], c);
                        }
                    } else {
                        const float sc = hbits2f(scb);
#pragma unroll
                        for (int i = 0; i < 32; i++) {
                            const int qv = (int)(int8_t)kbyte(b * 34 + 2 + i);
                            const float kvv = h2f(f2h((float)qv * sc));
                            c = __builtin_fmaf(h2f(qs[b * 32 + i]), kvv, c);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < HD / 2; i++) {
                    const half2_t k2 = __builtin_bit_cast(half2_t, kreg[i]);
                    c = __builtin_fmaf(h2f(qs[2 * i]), (float)k2[0], c);
                    c = __builtin_fmaf(h2f(qs[2 * i + 1]), (float)k2[1], c);
                }
            }
        }
        half_t s = f2h(alpha * c);
        if (P.alibi) { float a = (float)j * mk; s = f2h(a + h2f(s)); }
        S[j] = s;
        lmax = fmaxf(lmax, P.kq_scale * h2f(s));
    }
    if (tr) P.trace[h * 8 + 3] = wall_clock64();
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    if (tr) P.trace[h * 8 + 4] = wall_clock64();
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
    if (tr) P.trace[h * 8 + 5] = wall_clock64();

    // ---- O = P.V : thread (sp, dg) accumulates keys j = sp + NSPLIT*i for its 8 dims
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = 0.0f;
    auto acc_v = [&](float pj, const u32x4 vv) {
        const half8_t v8 = __builtin_bit_cast(half8_t, vv);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = __builtin_fmaf(pj, (float)v8[e], o[e]);
    };
    auto acc_new = [&](float pj) {
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = __builtin_fmaf(pj, h2f(vn[dg * 8 + e]), o[e]);
    };
    // 8 codes (four 2-byte pieces) of one Q8 V block times its scale, accumulated in element order
    auto acc_q8w = [&](float pj, uint16_t scb, uint16_t c0, uint16_t c1, uint16_t c2, uint16_t c3) {
        const half_t sch = __builtin_bit_cast(half_t, scb);
        const half2_t sc2 = {sch, sch};
        half2_t v01, v23, v45, v67;
        q8x4_dequant_h((uint32_t)c0 | ((uint32_t)c1 << 16), sc2, v01, v23);
        q8x4_dequant_h((uint32_t)c2 | ((uint32_t)c3 << 16), sc2, v45, v67);
        o[0] = __builtin_fmaf(pj, (float)v01[0], o[0]); o[1] = __builtin_fmaf(pj, (float)v01[1], o[1]);
        o[2] = __builtin_fmaf(pj, (float)v23[0], o[2]); o[3] = __builtin_fmaf(pj, (float)v23[1], o[3]);
        o[4] = __builtin_fmaf(pj, (float)v45[0], o[4]); o[5] = __builtin_fmaf(pj, (float)v45[1], o[5]);
        o[6] = __builtin_fmaf(pj, (float)v67[0], o[6]); o[7] = __builtin_fmaf(pj, (float)v67[1], o[7]);
    };
    auto acc_q8 = [&](float pj, int j) {
        const auto *blk = IFA_GP(uint16_t, pvc + (size_t)j * row_bytes + head_off + (size_t)(dg / 4) * 34);
        typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
        const u32x2_a2 cw = *IFA_GP(u32x2_a2, blk + 1 + (dg % 4) * 4);      // (one unaligned 8-byte request: see the prefetch above)
        acc_q8w(pj, blk[0], (uint16_t)(cw[0] & 0xFFFFu), (uint16_t)(cw[0] >> 16), (uint16_t)(cw[1] & 0xFFFFu), (uint16_t)(cw[1] >> 16));
    };
#pragma unroll
    for (int i = 0; i < VPRE; i++) {        // first 256 keys: V rows already in registers (static indexing)
        const int j = sp + NSPLIT * i;
        if (j < n_ctx && vact) {
            const float pj = h2f(S[j]);
            if (j == pos) acc_new(pj);
            else if constexpr (Q8) acc_q8w(pj, vq[i][0], vq[i][1], vq[i][2], vq[i][3], vq[i][4]);
            else acc_v(pj, vreg[i]);
        }
    }
    for (int j = vact ? sp + NSPLIT * VPRE : n_ctx; j < n_ctx; j += NSPLIT) {
        const float pj = h2f(S[j]);
        if (j == pos) acc_new(pj);
        else if constexpr (Q8) acc_q8(pj, j);
        else acc_v(pj, IFA_GP(u32x4, pvc + (size_t)j * row_bytes + head_off)[dg]);
    }
    if (vact) {
#pragma unroll
        for (int e = 0; e < 8; e++) opart[sp * HD + dg * 8 + e] = o[e];
    }
    if (tr) P.trace[h * 8 + 6] = wall_clock64();
    __syncthreads();
    if (tid < HD) {
        float acc = opart[tid];
        for (int s2 = 1; s2 < NSPLIT; s2++) acc = acc + opart[s2 * HD + tid];
        const half_t yh = f2h(acc);
        outp[(size_t)h * HD + tid] = yh;
        if constexpr (HD % 32 == 0 && !BATCH) { if (P.xq) dec_attn_emit_q8<HD>(P.xq, P.heads * HD, h, tid, yh); }
    }
    if (tr) P.trace[h * 8 + 7] = wall_clock64();
}

// ------------------------------------------------------------ long contexts: keys split over workgroups
// k_dec_attn keeps one workgroup per head, which is latency-optimal for a few hundred keys but reads the whole
// K/V history of a head through ONE compute unit (4K keys: ~190 us per layer).  Past a threshold the decode step
// uses three kernels instead, with the SAME rounding points (S and P are half, global max and sum):
//   k_dec_attn_scores  (head, split): RoPE, KV store of the new row, S_j = half(alpha q.k_j) for its keys -> workspace,
//                                      local maximum
//   k_dec_attn_pv      (head, split): global max, the full-row sum of exp (recomputed per split: a few thousand
//                                      expf), P_j = half(half(e_j) * 1/sum) for its keys, partial P.V in fp32
//   k_dec_attn_combine (head)       : sum of the partial outputs in split order -> half
constexpr int DEC_ATTN_MAX_SPLITS = 32;      // splits per head are chosen per decode call from the context it will reach (8 / 16 / 32)

struct DecAttnSplitWs {
    half_t *S;        // [heads][max_ctx]
    float *lmax;      // [heads][nsplits]
    float *opart;     // [heads][nsplits][head_dim]
    int nsplits;      // <= DEC_ATTN_MAX_SPLITS
};

__device__ __forceinline__ void dec_split_range(int n_ctx, int nsplits, int s, int &j0, int &j1)
{
    const int chunk = ((n_ctx + nsplits - 1) / nsplits + 63) / 64 * 64;
    j0 = min(s * chunk, n_ctx); j1 = min(j0 + chunk, n_ctx);
}

template <int HD, bool Q8>
__global__ void __launch_bounds__(256) k_dec_attn_scores(const DecAttnParams P, const DecAttnSplitWs ws)
{
    __shared__ __attribute__((aligned(16))) half_t qs[HD];
    __shared__ __attribute__((aligned(16))) half_t kn[HD];
    __shared__ __attribute__((aligned(16))) half_t vn[HD];
    __shared__ float red[4];
    const int h = blockIdx.x, sidx = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = *(const __attribute__((address_space(4))) int *)(P.state + 1), n_ctx = pos + 1;      // (scalar load: see k_dec_attn)
    int j0, j1; dec_split_range(n_ctx, ws.nsplits, sidx, j0, j1);
    const int group = P.heads / P.kv_heads, kvh = h / group;
    const bool has_new = pos >= j0 && pos < j1;             // this split owns the new token's row
    const bool writer = has_new && (h % group) == 0;
    const int kv_dim = P.kv_heads * HD;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const size_t head_off = Q8 ? (size_t)((kvh * HD) / 32) * 34 : (size_t)kvh * HD * 2;
    for (int d = tid; d < HD; d += 256) {
        qs[d] = P.q[(size_t)h * HD + d];
        kn[d] = P.k_new[(size_t)kvh * HD + d];
        vn[d] = P.v_new[(size_t)kvh * HD + d];
    }
    __syncthreads();
    if (P.rope_order != 0) {
        if (tid < HD) {
            const int c = tid < HD / 2 ? tid : tid - HD / 2;
            rope_apply(tid < HD / 2 ? qs : kn, c, P.rope_tab[2 * c], P.rope_tab[2 * c + 1], P.rope_order, P.rope_cols);
        }
        __syncthreads();
    }
    if (has_new) {      // KV store of the new row (and its Q8 round trip), as in k_dec_attn
        if constexpr (Q8) {
            constexpr int NB = HD / 32;
            for (int b = wave; b < 2 * NB; b += 4) {
                half_t *src = b < NB ? kn : vn;
                const int bb = b < NB ? b : b - NB;
                if (lane < 32) {
                    const float val = h2f(src[bb * 32 + lane]);
                    float mx = fabsf(val);
#pragma unroll
                    for (int m2 = 16; m2 > 0; m2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m2, 32));
                    const float sc = mx / 127;
                    int qv = sc <= 0.000001f ? 0 : (int)roundf(val / sc);
                    qv = min(max(qv, -128), 127);
                    const half_t sch = f2h(sc);
                    if (writer) {
                        uint8_t *cache = b < NB ? P.kcache : P.vcache;
                        uint8_t *blk = cache + (size_t)pos * row_bytes + head_off + (size_t)bb * 34;
                        blk[2 + lane] = (uint8_t)(int8_t)qv;
                        if (lane == 0) *reinterpret_cast<uint16_t *>(blk) = __builtin_bit_cast(uint16_t, sch);
                    }
                    src[bb * 32 + lane] = f2h((float)qv * h2f(sch));
                }
            }
            __syncthreads();
        } else if (writer && tid < HD) {
            reinterpret_cast<half_t *>(P.kcache + (size_t)pos * row_bytes + head_off)[tid] = kn[tid];
            reinterpret_cast<half_t *>(P.vcache + (size_t)pos * row_bytes + head_off)[tid] = vn[tid];
        }
    }
    const float alpha = 1.0f / sqrtf((float)HD) / P.kq_scale;
    const float mk = P.alibi ? alibi_slope(h + P.alibi_base, P.alibi_total) : 0.0f;
    float lmax = -INFINITY;
    // F16 cache: a lane's key row (HD halfs) is requested COALESCED -- a wave request covers 64 / (HD/8) whole rows -- and
    // turned through LDS so that every lane then holds its own key's row: one key per lane with its own 16-byte requests
    // touched 64 rows per request (14 us per layer at 4096 keys).  The dot product below is unchanged (fp32 fma in d order).
    constexpr int KCH = HD / 8;                                   // 16-byte chunks per row
    constexpr int KROWB = HD * 2 + 16;                            // bytes per staged row (+16: conflict-free row reads)
    extern __shared__ __attribute__((aligned(16))) char kst[];    // [256][KROWB] (F16 cache only: dec_attn_scores_smem)
    for (int jb = j0; jb < j1; jb += 256) {
        const int j = jb + tid;
        u32x4 krow[Q8 ? 1 : KCH];
        if constexpr (!Q8) {
            u32x4 kin[KCH];
#pragma unroll
            for (int i2 = 0; i2 < KCH; i2++) {                    // piece idx = tid + 256 i2: row idx / KCH, chunk idx % KCH
                const int idx = tid + 256 * i2;
                const int jr = min(jb + idx / KCH, j1 - 1);
                kin[i2] = reinterpret_cast<const u32x4 *>(P.kcache + (size_t)jr * row_bytes + head_off)[idx % KCH];
            }
            __syncthreads();                                      // the previous pass's rows have been read
#pragma unroll
            for (int i2 = 0; i2 < KCH; i2++) {
                const int idx = tid + 256 * i2;
                *reinterpret_cast<u32x4 *>(kst + (size_t)(idx / KCH) * KROWB + (size_t)(idx % KCH) * 16) = kin[i2];
            }
            __syncthreads();
#pragma unroll
            for (int i2 = 0; i2 < KCH; i2++) krow[i2] = *reinterpret_cast<const u32x4 *>(kst + (size_t)tid * KROWB + (size_t)i2 * 16);
        }
        if (j >= j1) continue;
        float c = 0.0f;
        if (j == pos) {
#pragma unroll 8
            for (int d = 0; d < HD; d++) c = __builtin_fmaf(h2f(qs[d]), h2f(kn[d]), c);
        } else {
            const uint8_t *rowp = P.kcache + (size_t)j * row_bytes + head_off;
            if constexpr (Q8) {
#pragma unroll
                for (int b = 0; b < HD / 32; b++) {
                    const uint16_t *p16 = reinterpret_cast<const uint16_t *>(rowp + b * 34);
                    const float sc = hbits2f(p16[0]);
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint32_t two = p16[1 + i];
                        const float k0 = h2f(f2h((float)(int)(int8_t)(two & 0xFF) * sc)), k1 = h2f(f2h((float)(int)(int8_t)(two >> 8) * sc));
                        c = __builtin_fmaf(h2f(qs[b * 32 + 2 * i]), k0, c);
                        c = __builtin_fmaf(h2f(qs[b * 32 + 2 * i + 1]), k1, c);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < HD / 8; i++) {
                    const half8_t k8 = __builtin_bit_cast(half8_t, krow[i]);
#pragma unroll
                    for (int e = 0; e < 8; e++) c = __builtin_fmaf(h2f(qs[8 * i + e]), (float)k8[e], c);
                }
            }
        }
        half_t sv = f2h(alpha * c);
        if (P.alibi) { float a = (float)j * mk; sv = f2h(a + h2f(sv)); }
        ws.S[(size_t)h * P.max_ctx + j] = sv;
        lmax = fmaxf(lmax, P.kq_scale * h2f(sv));
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    if (tid == 0) ws.lmax[h * ws.nsplits + sidx] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <int HD, bool Q8>
__global__ void __launch_bounds__(256) k_dec_attn_pv(const DecAttnParams P, const DecAttnSplitWs ws)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DG = HD / 8, NSPLIT = 256 / DG;
    float *red = reinterpret_cast<float *>(smem);                    // [8]
    float *opart = red + 8;                                          // [NSPLIT][HD]
    half_t *Pl = reinterpret_cast<half_t *>(opart + NSPLIT * HD);    // this split's probabilities
    const int h = blockIdx.x, sidx = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = *(const __attribute__((address_space(4))) int *)(P.state + 1), n_ctx = pos + 1;      // (scalar load: see k_dec_attn)
    int j0, j1; dec_split_range(n_ctx, ws.nsplits, sidx, j0, j1);
    const int group = P.heads / P.kv_heads, kvh = h / group;
    const int kv_dim = P.kv_heads * HD;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const size_t head_off = Q8 ? (size_t)((kvh * HD) / 32) * 34 : (size_t)kvh * HD * 2;
    const half_t *Sg = ws.S + (size_t)h * P.max_ctx;
    float mx = -INFINITY;
    for (int s2 = 0; s2 < ws.nsplits; s2++) mx = fmaxf(mx, ws.lmax[h * ws.nsplits + s2]);
    // the full-row sum, in the same order as the one-workgroup kernel (strided by 256, wave tree, 4 waves)
    float lsum = 0.0f;
    for (int j = tid; j < n_ctx; j += 256) lsum += expf(P.kq_scale * h2f(Sg[j]) - mx);
    lsum = wave_sum(lsum);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / (((red[0] + red[1]) + red[2]) + red[3]);
    for (int j = j0 + tid; j < j1; j += 256) {
        const half_t eh = f2h(expf(P.kq_scale * h2f(Sg[j]) - mx));
        Pl[j - j0] = f2h(h2f(eh) * inv);
    }
    __syncthreads();
    const int dg = tid % DG, sp = tid / DG;
    const bool vact = (256 % DG == 0) || sp < NSPLIT;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = 0.0f;
    for (int j = vact ? j0 + sp : j1; j < j1; j += (Q8 ? 1 : 4) * NSPLIT) {
        const float pj = h2f(Pl[j - j0]);
        if constexpr (Q8) {
            const uint8_t *blk = P.vcache + (size_t)j * row_bytes + head_off + (size_t)(dg / 4) * 34;
            const uint16_t *p16 = reinterpret_cast<const uint16_t *>(blk);
            const float sc = hbits2f(p16[0]);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t two = p16[1 + (dg % 4) * 4 + e];
                o[2 * e] = __builtin_fmaf(pj, h2f(f2h((float)(int)(int8_t)(two & 0xFF) * sc)), o[2 * e]);
                o[2 * e + 1] = __builtin_fmaf(pj, h2f(f2h((float)(int)(int8_t)(two >> 8) * sc)), o[2 * e + 1]);
            }
        } else {
            // four rows of this thread's key sequence are requested before the first is multiplied (one request in flight
            // per thread left the split kernels at 1.6 TB/s); rows past the split are clamped and skipped -- same order of
            // accumulation as a plain loop
            u32x4 vr[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                vr[u] = reinterpret_cast<const u32x4 *>(P.vcache + (size_t)min(j + u * NSPLIT, j1 - 1) * row_bytes + head_off)[dg];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (j + u * NSPLIT >= j1) break;
                const float pu = h2f(Pl[j + u * NSPLIT - j0]);
                const half8_t v8 = __builtin_bit_cast(half8_t, vr[u]);
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = __builtin_fmaf(pu, (float)v8[e], o[e]);
            }
        }
    }
    if (vact) {
#pragma unroll
        for (int e = 0; e < 8; e++) opart[sp * HD + dg * 8 + e] = o[e];
    }
    __syncthreads();
    if (tid < HD) {
        float acc = opart[tid];
        for (int s2 = 1; s2 < NSPLIT; s2++) acc = acc + opart[s2 * HD + tid];
        ws.opart[((size_t)h * ws.nsplits + sidx) * HD + tid] = acc;
    }
}

template <int HD>
__global__ void __launch_bounds__(HD) k_dec_attn_combine(const DecAttnSplitWs ws, half_t *__restrict__ out, int8_t *xq, int heads)
{
    const int h = blockIdx.x, d = threadIdx.x;
    const float *p = ws.opart + (size_t)h * ws.nsplits * HD + d;
    float acc = p[0];
    for (int s2 = 1; s2 < ws.nsplits; s2++) acc = acc + p[(size_t)s2 * HD];
    const half_t yh = f2h(acc);
    out[(size_t)h * HD + d] = yh;
    if constexpr (HD % 32 == 0) { if (xq) dec_attn_emit_q8<HD>(xq, heads * HD, h, d, yh); }
}

__host__ __device__ inline size_t dec_attn_scores_smem(int head_dim, bool q8) { return q8 ? 16 : (size_t)256 * (head_dim * 2 + 16); }

__host__ __device__ inline size_t dec_attn_pv_smem(int head_dim, int max_ctx, int nsplits = 8)
{
    const size_t nsplit = 256 / (head_dim / 8);
    const size_t chunk = (((size_t)max_ctx + nsplits - 1) / nsplits + 63) / 64 * 64;
    return 8 * 4 + nsplit * head_dim * 4 + chunk * 2 + 16;
}

__host__ __device__ inline size_t dec_attn_smem(int head_dim, int max_ctx)
{
    const size_t nsplit = 256 / (head_dim / 8);
    return (size_t)head_dim * 3 * 2 + 16 * 4 + nsplit * head_dim * 4 + (((size_t)max_ctx * 2 + 15) & ~(size_t)15) + 16;
}

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
