// ifa_decode_persist.h -- the batch-1 decode step's layers as ONE persistent launch.
//
// The five-launch layer of ifa_decode_kernels.h reads its 126 MB (Llama-2-7B Q4) with no re-reads, but every launch
// pays ~4.5 us of boundary + first-request + first-byte latency during which HBM idles.  Weights
// never depend on activations, so here the weight stream is decoupled from the dependency chain:
//
//   * one workgroup of 4 waves per CU, resident for all layers of the token;
//   * wave 3 = LOADER: streams this CU's share of wq|wk|wv -> wo -> w1/w3 -> w2 -> next layer ... into a 112 KiB LDS
//     ring with direct-to-LDS loads (global_load_lds_dwordx4, non-temporal), 32 KiB in flight, never waiting for an
//     activation: while the consumers sit at a hand-off it runs ahead until the ring is full;
//   * waves 0-2 = CONSUMERS: per op they gather the op's input vector from the other CUs, (normalise and) quantise it
//     to Q8 blocks exactly like the five-launch prologue, keep their slice in registers, and reduce rows out of the
//     ring with the SAME per-lane block order, dot code and wave reduction as k_dec_gemv -- results are bit-identical
//     to the five-launch path, which stays as fallback and comparator (tests/test_gpu_persist.py);
//   * hand-offs between CUs are 8-byte {tag, data} granules written by one write-through (sc1) store and polled with
//     sc1 loads (MI355X_MICROARCH.md "Persistent kernels" price list, Guideline 16 form R2): no flags, no fences.
//     tag = layer * 8 + edge + 1; the host zeroes the granule arena before every launch;
//   * attention of head h runs on one CU (spread over the XCDs) with the arithmetic order of k_dec_attn (256 virtual
//     threads walked by the 3 consumer waves); it leaves its output quantised, like k_dec_attn does for Wo.
//
// Reference sequence preserved: src/transformer/inference_worker.cc:762-981 (one decoder layer), :1116-1312
// (self-attention), kernels src/kernels/gemv.h:1499-1709 (int8 GEMV terms), tensor_quant.h:44-82 (Alg2 quantiser).
//
// Every wait is bounded: a spin that exceeds P.timeout_ticks records an error code in P.err, raises the workgroup's
// abort flag and from then on no wait blocks, so a broken hand-off ends the launch in milliseconds instead of
// hanging the GPU; the host turns a non-zero P.err into an error return (ifa_model_decode).
#pragma once
#include "ifa_decode_kernels.h"

namespace ifa {

typedef __attribute__((address_space(1))) unsigned long long ps_gu64;
typedef __attribute__((address_space(1))) unsigned int ps_gu32;
typedef __attribute__((address_space(3))) void ps_lds_t;
typedef const __attribute__((address_space(1))) void ps_glb_t;

constexpr int PS_NC = 3;                         // consumer waves per workgroup
constexpr int PS_THREADS = 64 * (PS_NC + 1);     // + the loader wave
constexpr uint32_t PS_RING = 112u * 1024u;       // LDS ring of the weight stream (7 x 16 KiB)
constexpr int PS_RB = 2;                         // rows (GLU: row pairs) of a consumer batch = one granule of output
constexpr int PS_MAX_CTX = 1024;                 // keys one CU handles per head; beyond, the host uses the five-launch path
constexpr int PS_RES = 512;                      // residual values a CU may own per op

#ifndef IFA_PS_INFLIGHT
#define IFA_PS_INFLIGHT 32                       // KiB of direct-to-LDS loads in flight per CU
#endif
#ifndef IFA_PS_THIN
#define IFA_PS_THIN 16                           // ... while a wave of this CU sweeps granules (price list: gather-pass)
#endif
#ifndef IFA_PS_NT
#define IFA_PS_NT 1                              // non-temporal policy on the weight stream (price list: nt-weights)
#endif

// control words (dword index into the ctl block)
enum { PS_C_FILLED = 0, PS_C_NEED = 1 /* 3 */, PS_C_BAR = 4, PS_C_ABORT = 5, PS_C_GATHER = 6, PS_C_PART = 16 /* 48 floats */, PS_C_WORDS = 128 };
enum { PS_E_X = 0, PS_E_QKV = 1, PS_E_ATT = 2, PS_E_A = 3, PS_E_ACT = 4 };
__host__ __device__ constexpr unsigned ps_epoch(int layer, int edge) { return (unsigned)(layer * 8 + edge + 1); }

// error codes: (phase << 8) | kind ; P.err[0] = code, [1] = workgroup, [2] = layer, [3] = wave
enum { PS_ERR_RING = 1, PS_ERR_BAR = 2, PS_ERR_GATHER = 3, PS_ERR_SPACE = 4, PS_ERR_HINT = 5 };

struct PsLayer {                 // one per layer, device memory, read with scalar loads
    // tiled rows (ifa_tiled.h) in the order the loader streams them: wq | wk | wv rows back to back in ONE buffer, and
    // w1 / w3 interleaved row by row (row 2r = w1 row r, row 2r + 1 = w3 row r), so that every op of a CU is one
    // contiguous byte range (copies made at load time: ifa_engine.hip, persist_build)
    const uint8_t *wqkv, *wo, *w13, *w2;
    const half_t *attn_norm, *attn_norm_b, *ffn_norm, *ffn_norm_b;
    const half_t *bq, *bk, *bv, *bo, *b1, *b3, *b2;
    uint8_t *kcache, *vcache;
};

struct PsParams {
    const PsLayer *layers;
    const half_t *x_in;          // input of layer `layer_begin` (plain F16, written by an earlier launch)
    half_t *x_out;               // output of layer `layer_end - 1` (plain F16)
    const int *state;            // state[1] = position of the new token
    const float *rope_tab;
    unsigned long long *g_x, *g_qkv, *g_att, *g_a, *g_act;      // granule arenas of the five edges
    unsigned *err;
    long long *trace;            // optional [workgroups][32] stamps of layer trace_layer
    half_t *dbg_att;             // optional F16 copy of the attention output (tests)
    int layer_begin, layer_end;
    int dim, ffn, heads, kv_heads;
    int nblk_a, nblk_b;          // weight blocks per row: dim-wide and ffn-wide matrices
    unsigned row_bytes_a, row_bytes_b;      // tiled row strides
    float eps, attn_norm_base, ffn_norm_base, kq_scale;
    int act_kind, rope_order, rope_cols, alibi, alibi_base, alibi_total;
    int trace_layer;
    unsigned timeout_ticks;      // 100 MHz ticks
};
static_assert(sizeof(PsParams) <= 256, "PsParams: keep the argument block within 256 bytes");

// balanced contiguous split of `total` items over ncu workgroups
__host__ __device__ inline void ps_part(int total, int ncu, int cu, int &first, int &count)
{
    const int base = total / ncu, rem = total % ncu;
    first = cu * base + (cu < rem ? cu : rem);
    count = base + (cu < rem ? 1 : 0);
}

// head -> workgroup: strided over the grid, offset so that consecutive heads land on different XCDs (block b runs on
// XCD b % 8, observed, used for speed only)
__host__ __device__ inline int ps_head_cu(int h, int heads, int ncu)
{
    const int stride = ncu / heads;
    return h * stride + (stride >= 8 ? (h & 7) : 0);
}

// per-CU geometry of a layer's four weight ops; identical in the loader and the consumers
struct PsGeom {
    int first[4], n[4];          // op 0: virtual q|k|v rows, 1: wo rows, 2: act rows (each = a w1 row and a w3 row), 3: w2 rows
    uint32_t off[4], layer_bytes;
};

__host__ __device__ inline PsGeom ps_geom(int dim, int ffn, int q_rows, int kv_rows, unsigned rb_a, unsigned rb_b, int ncu, int cu)
{
    PsGeom g;
    int f, c;
    ps_part((q_rows + 2 * kv_rows) / 2, ncu, cu, f, c); g.first[0] = 2 * f; g.n[0] = 2 * c;
    ps_part(dim / 2, ncu, cu, f, c); g.first[1] = 2 * f; g.n[1] = 2 * c;
    ps_part(ffn / 2, ncu, cu, f, c); g.first[2] = 2 * f; g.n[2] = 2 * c;
    ps_part(dim / 2, ncu, cu, f, c); g.first[3] = 2 * f; g.n[3] = 2 * c;
    uint32_t o = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t bytes = (uint32_t)g.n[i] * (i == 2 ? 2u : 1u) * (i == 3 ? rb_b : rb_a);
        g.off[i] = o;
        o += (bytes + 1023u) & ~1023u;
    }
    g.layer_bytes = o;
    return g;
}

__host__ __device__ inline size_t ps_attn_scratch_bytes(int hd)
{
    return (size_t)hd * 7 + 64 + (size_t)(256 / (hd / 8)) * hd * 4 + (size_t)PS_MAX_CTX * 2 + 64;
}
__host__ __device__ inline size_t ps_img_bytes(int maxcols) { return ((size_t)maxcols + (size_t)maxcols / 32 * 8 + 15) / 16 * 16; }
__host__ __device__ inline size_t ps_stage_bytes(int maxcols, int hd)
{
    const size_t a = (size_t)maxcols * 2, b = ps_attn_scratch_bytes(hd);
    return ((a > b ? a : b) + 15) / 16 * 16;
}
constexpr size_t PS_CTL_OFF = PS_RING, PS_RES_OFF = PS_RING + PS_C_WORDS * 4, PS_IMG_OFF = PS_RES_OFF + 2 * PS_RES * 2;
__host__ __device__ inline size_t ps_lds_bytes(int maxcols, int hd) { return PS_IMG_OFF + ps_img_bytes(maxcols) + ps_stage_bytes(maxcols, hd); }

// ------------------------------------------------------------------ workgroup-local plumbing
struct PsCtx {
    char *smem;
    unsigned *ctl;
    int lane, w, cu, ncu;
    unsigned bar_gen, filled_seen, timeout;
    unsigned *err;
    long long *trace;
    int cur_layer;
};

__device__ __forceinline__ unsigned ps_lds_ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ps_lds_st(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ps_lds_add(unsigned *p, unsigned v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define PS_CB() asm volatile("" ::: "memory")

__device__ __forceinline__ void ps_stamp(const PsCtx &c, int idx)
{
    if (c.trace && c.lane == 0) c.trace[idx] = wall_clock64();
}

__device__ __forceinline__ void ps_fail(PsCtx &c, unsigned code)
{
    ps_lds_st(c.ctl + PS_C_ABORT, 1u);
    if (c.lane == 0) {
        if (atomicCAS(c.err, 0u, code) == 0u) { c.err[1] = (unsigned)c.cu; c.err[2] = (unsigned)c.cur_layer; c.err[3] = (unsigned)c.w; }
    }
}

// inside a spin loop: true = stop waiting (this workgroup or another one gave up)
__device__ __forceinline__ bool ps_expired(PsCtx &c, long long t0, unsigned code)
{
    if (ps_lds_ld(c.ctl + PS_C_ABORT)) return true;
    const long long dt = wall_clock64() - t0;
    if (dt > (long long)c.timeout) { ps_fail(c, code); return true; }
    if (dt > 2000 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {      // someone else failed (looked at after 20 us of waiting)
        ps_lds_st(c.ctl + PS_C_ABORT, 1u);
        return true;
    }
    return false;
}
__device__ __forceinline__ bool ps_aborted(const PsCtx &c) { return ps_lds_ld(c.ctl + PS_C_ABORT) != 0u; }

// barrier over the consumer waves (the loader never joins a barrier: it would stop streaming)
__device__ __forceinline__ void ps_cbar(PsCtx &c, unsigned code)
{
    c.bar_gen += PS_NC;
    PS_CB();
    if (c.lane == 0) ps_lds_add(c.ctl + PS_C_BAR, 1u);
    const long long t0 = wall_clock64();
    while ((int)(ps_lds_ld(c.ctl + PS_C_BAR) - c.bar_gen) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (ps_expired(c, t0, (code << 8) | PS_ERR_BAR)) break;
    }
    PS_CB();
}

// wait until the loader has landed the stream up to byte position `end`
__device__ __forceinline__ void ps_wait_filled(PsCtx &c, uint32_t end, unsigned code)
{
    if ((int)(c.filled_seen - end) >= 0) return;
    const long long t0 = wall_clock64();
    for (;;) {
        c.filled_seen = ps_lds_ld(c.ctl + PS_C_FILLED);
        if ((int)(c.filled_seen - end) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
        if (ps_expired(c, t0, (code << 8) | PS_ERR_RING)) break;
    }
    PS_CB();
}

__device__ __forceinline__ void ps_publish(unsigned long long *g, int idx, unsigned epoch, uint32_t value)
{
    __hip_atomic_store(((ps_gu64 *)g) + idx, ((unsigned long long)epoch << 32) | (unsigned long long)value,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ps_peek(const unsigned long long *g, int idx)
{
    return __hip_atomic_load(((const ps_gu64 *)g) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// This wave sweeps granules [first, first + n) of `g` until every tag equals `epoch` and writes the 32-bit payloads to
// dst[granule index] (LDS dwords).  A cheap hint poll (one granule per lane: the last of each 1/64 slice) comes first so
// that 256 CUs waiting for one edge do not sweep the whole vector at the fabric every microsecond.
template <int MAXL>
__device__ __forceinline__ void ps_gather(PsCtx &c, const unsigned long long *g, int first, int n, unsigned epoch, uint32_t *dst, unsigned code)
{
    if (n <= 0) return;
    if (c.lane == 0) ps_lds_add(c.ctl + PS_C_GATHER, 1u);
    const long long t0 = wall_clock64();
    {
        const int hi = first + max(0, ((c.lane + 1) * n) / 64 - 1);
        for (;;) {
            const unsigned long long x = ps_peek(g, hi);
            if (__all((unsigned)(x >> 32) == epoch)) break;
            __builtin_amdgcn_s_sleep(2);
            if (ps_expired(c, t0, (code << 8) | PS_ERR_HINT)) break;
        }
    }
    for (int base = 0; base < n; base += 64 * MAXL) {
        for (;;) {
            uint32_t v[MAXL];
            bool ok = true;
#pragma unroll
            for (int k = 0; k < MAXL; k++) {
                const int idx = min(base + c.lane + 64 * k, n - 1);
                const unsigned long long x = ps_peek(g, first + idx);
                v[k] = (uint32_t)x;
                ok &= (unsigned)(x >> 32) == epoch;
            }
            if (__all(ok)) {
#pragma unroll
                for (int k = 0; k < MAXL; k++) {
                    const int idx = base + c.lane + 64 * k;
                    if (idx < n) dst[first + idx] = v[k];
                }
                break;
            }
            if (ps_expired(c, t0, (code << 8) | PS_ERR_GATHER)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (c.lane == 0) ps_lds_add(c.ctl + PS_C_GATHER, 0xFFFFFFFFu);
}

// LDS image of a quantised activation (same three arrays as XLds / XqImage, contiguous: codes | scale | xsum)
struct PsImg { int8_t *codes; float *scale; float *xsum; };
__device__ __forceinline__ PsImg ps_img(char *base, int cols)
{
    PsImg q;
    q.codes = reinterpret_cast<int8_t *>(base);
    q.scale = reinterpret_cast<float *>(base + cols);
    q.xsum = q.scale + cols / 32;
    return q;
}

// [RMS-normalise and] quantise the staged F16 vector into the image: the arithmetic of XPre::finish (same chunk -> lane
// mapping of the canonical RMS order, same quad-local Q8_B32T2 quantiser), the 64-chunk groups dealt to the consumer waves.
template <bool NORM, int MAXG>
__device__ __forceinline__ void ps_quantize(PsCtx &c, const half_t *stage, int cols, const half_t *nw, const half_t *nb, float multi_base,
                                            float eps, const PsImg &L, unsigned code)
{
    const int chunks = cols >> 3, ngroups = (chunks + 63) >> 6;
    float *part = reinterpret_cast<float *>(c.ctl + PS_C_PART);
    half8_t wv[NORM ? MAXG : 1], bv[NORM ? MAXG : 1];
    float scale = 1.0f;
    if constexpr (NORM) {
#pragma unroll
        for (int gi = 0; gi < MAXG; gi++) {
            const int ch = 64 * (c.w + PS_NC * gi) + c.lane;
            if (ch < chunks) {
                if (nw) wv[gi] = *reinterpret_cast<const half8_t *>(nw + (size_t)ch * 8);
                if (nb) bv[gi] = *reinterpret_cast<const half8_t *>(nb + (size_t)ch * 8);
            }
        }
#pragma unroll
        for (int gi = 0; gi < MAXG; gi++) {
            const int grp = c.w + PS_NC * gi;
            if (grp < ngroups) {
                const int ch = 64 * grp + c.lane;
                half8_t v8;
#pragma unroll
                for (int i = 0; i < 8; i++) v8[i] = (half_t)0;
                if (ch < chunks) v8 = *reinterpret_cast<const half8_t *>(stage + (size_t)ch * 8);
                const float pg = wave_sum(rms_chunk_sq(v8));
                if (c.lane == 0) part[grp] = pg;
            }
        }
        ps_cbar(c, code);
        scale = rms_scale_of(rms_total(part, ngroups), cols, eps);
    }
#pragma unroll
    for (int gi = 0; gi < MAXG; gi++) {
        const int ch = 64 * (c.w + PS_NC * gi) + c.lane;
        if (ch >= chunks) continue;          // whole quads (4 lanes = one block) are in or out together
        const half8_t xv = *reinterpret_cast<const half8_t *>(stage + (size_t)ch * 8);
        float v[8];
        if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float t = (float)xv[i] * scale;
                if (nw) {
                    float m = multi_base + (float)wv[gi][i];
                    t = t * m;
                    if (nb) t = t + (float)bv[gi][i];
                }
                v[i] = h2f(f2h(t));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (float)xv[i];
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
        *reinterpret_cast<u32x2 *>(L.codes + (size_t)ch * 8) = packed;
        if ((ch & 3) == 0) {
            L.scale[ch >> 2] = h2f(f2h(qs));
            L.xsum[ch >> 2] = (float)s;
        }
    }
    ps_cbar(c, code);
}

// Rows of one op out of the ring: batch k = PS_RB rows (x NM matrices) = units [k * UB, (k + 1) * UB) of this CU's stream
// segment; batches are dealt to the consumer waves round-robin.  Same lane -> block mapping, dot() and wave_sum() as
// k_dec_gemv; epi(k, a0, a1) is called by every lane with lane i < PS_RB holding row i's sums.
template <int DT, int NJ, int NM, class Epi>
__device__ __forceinline__ void ps_gemv(PsCtx &c, const typename DecFmt<DT, NJ>::X &X, int nblk, uint32_t row_bytes, uint32_t op_pos,
                                        uint32_t op_end, int n_units, unsigned code, Epi &&epi)
{
    using Fmt = DecFmt<DT, NJ>;
    constexpr int UB = PS_RB * NM;
    const int nb = n_units / UB;
    const uint32_t bbytes = (uint32_t)UB * row_bytes;
    for (int k = c.w; k < nb; k += PS_NC) {
        const uint32_t pos0 = op_pos + (uint32_t)k * bbytes;
        ps_wait_filled(c, pos0 + bbytes, code);
        typename Fmt::W wr[NM][PS_RB];
#pragma unroll
        for (int i = 0; i < PS_RB; i++)
#pragma unroll
            for (int m = 0; m < NM; m++) {
                const WSrcLdsRing<PS_RING> src = {c.smem, (pos0 + (uint32_t)(i * NM + m) * row_bytes) % PS_RING};
                wr[m][i].load_src(src, nblk, c.lane);
            }
        // LDS executes a wave's instructions in order: this store is behind the reads above, so the loader sees the
        // space as free only after they were served
        PS_CB();
        if (c.lane == 0) ps_lds_st(c.ctl + PS_C_NEED + c.w, k + PS_NC < nb ? pos0 + (uint32_t)PS_NC * bbytes : op_end);
        PS_CB();
        float a[NM][PS_RB];
#pragma unroll
        for (int i = 0; i < PS_RB; i++)
#pragma unroll
            for (int m = 0; m < NM; m++) a[m][i] = wr[m][i].dot(X);
#pragma unroll
        for (int i = 0; i < PS_RB; i++)
#pragma unroll
            for (int m = 0; m < NM; m++) a[m][i] = wave_sum(a[m][i]);
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int i = 0; i < PS_RB; i++) {
            if (c.lane == i) { a0 = a[0][i]; if constexpr (NM == 2) a1 = a[1][i]; }
        }
        epi(k, a0, a1);
    }
    if (nb <= c.w && c.lane == 0) ps_lds_st(c.ctl + PS_C_NEED + c.w, op_end);
}

// two half results of lanes 0 / 1 -> one granule, stored by lane 0
__device__ __forceinline__ void ps_publish_pair(const PsCtx &c, unsigned long long *g, int idx, unsigned epoch, half_t y)
{
    const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, y);
    const uint32_t other = (uint32_t)__shfl(mine, 1);
    if (c.lane == 0) ps_publish(g, idx, epoch, mine | (other << 16));
}

// ------------------------------------------------------------------ attention of one head on this CU
// The arithmetic and its ORDER are k_dec_attn's (scores: fp32 fma chain in d order; S and P rounded to half; maximum and
// sum per 64-key "wave" of a 256-thread workgroup, combined in wave order; P.V per (key residue, 8-dim group) thread,
// partials added in residue order): the 256 threads of that kernel are walked as virtual threads by the 192 lanes here.
template <int HD, bool Q8>
__device__ __forceinline__ void ps_attention(PsCtx &c, const PsParams &P, const PsLayer &ly, int layer, int h, int pos, char *scratch)
{
    constexpr int DG = HD / 8, NSPLIT = 256 / DG, VPRE = 256 / NSPLIT;
    static_assert(HD == 64 || HD == 128, "persistent attention: head_dim 64 or 128");
    half_t *qs = reinterpret_cast<half_t *>(scratch);
    half_t *kn = qs + HD, *vn = kn + HD;
    float *red = reinterpret_cast<float *>(vn + HD);                  // [16]
    int8_t *cod = reinterpret_cast<int8_t *>(red + 16);               // [HD]
    float *opart = reinterpret_cast<float *>(cod + HD);               // [NSPLIT][HD]
    half_t *S = reinterpret_cast<half_t *>(opart + NSPLIT * HD);      // [n_ctx]
    const int lane = c.lane, w = c.w, tid = w * 64 + lane;
    const int n_ctx = pos + 1;
    const int group = P.heads / P.kv_heads, kvh = h / group;
    const bool writer = (h % group) == 0;
    const int q_rows = P.heads * HD, kv_dim = P.kv_heads * HD;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const size_t head_off = Q8 ? (size_t)((kvh * HD) / 32) * 34 : (size_t)kvh * HD * 2;
    const uint8_t *pkc = ly.kcache, *pvc = ly.vcache;

    constexpr int KBYTES = (HD / 32) * 34;
    constexpr int KALIGN = HD == 128 ? 8 : 4;
    uint32_t kreg[Q8 ? 1 : HD / 2];
    uint32_t kq32[Q8 ? KBYTES / 4 : 1];
    auto load_k = [&](int j) {
        const uint8_t *rowp = pkc + (size_t)j * row_bytes + head_off;
        if constexpr (Q8) {
            if constexpr (KALIGN == 8) {
#pragma unroll
                for (int i = 0; i < KBYTES / 8; i++) {
                    const u32x2 t = reinterpret_cast<const u32x2 *>(rowp)[i];
                    kq32[2 * i] = t[0]; kq32[2 * i + 1] = t[1];
                }
            } else {
#pragma unroll
                for (int i = 0; i < KBYTES / 4; i++) kq32[i] = reinterpret_cast<const uint32_t *>(rowp)[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < HD / 8; i++) {
                const u32x4 t = reinterpret_cast<const u32x4 *>(rowp)[i];
                kreg[4 * i] = t[0]; kreg[4 * i + 1] = t[1]; kreg[4 * i + 2] = t[2]; kreg[4 * i + 3] = t[3];
            }
        }
    };
    auto kbyte = [&](int B) -> uint32_t { return (kq32[B >> 2] >> (8 * (B & 3))) & 0xFFu; };

    // requests first: this lane's first key row and first V rows (the caches hold >= DEC_ATTN_MIN_ROWS rows: no clamp)
    load_k(tid);
    const int dg0 = tid % DG, sp0 = tid / DG;
    u32x4 vreg[Q8 ? 1 : VPRE];
    uint16_t vq[Q8 ? VPRE : 1][5];
#pragma unroll
    for (int i = 0; i < VPRE; i++) {
        const int j = min(sp0 + NSPLIT * i, DEC_ATTN_MIN_ROWS - 1);
        if constexpr (!Q8) {
            vreg[i] = reinterpret_cast<const u32x4 *>(pvc + (size_t)j * row_bytes + head_off)[dg0];
        } else {
            const uint16_t *blk = reinterpret_cast<const uint16_t *>(pvc + (size_t)j * row_bytes + head_off + (size_t)(dg0 / 4) * 34);
            vq[i][0] = blk[0];
#pragma unroll
            for (int e = 0; e < 4; e++) vq[i][1 + e] = blk[1 + (dg0 % 4) * 4 + e];
        }
    }
    float rope_cs = 1.0f, rope_sn = 0.0f;
    if (P.rope_order != 0) {
        const int cc = min(tid < HD / 2 ? tid : tid - HD / 2, HD / 2 - 1);
        rope_cs = P.rope_tab[2 * cc]; rope_sn = P.rope_tab[2 * cc + 1];
    }
    // the new token's q | k | v of this head: three granule runs of HD/2 each, gathered by wave 0
    if (w == 0) {
        const unsigned ep = ps_epoch(layer, PS_E_QKV);
        const int l2 = min(lane, HD / 2 - 1);
        const int iq = (h * HD) / 2 + l2, ik = (q_rows + kvh * HD) / 2 + l2, iv = (q_rows + kv_dim + kvh * HD) / 2 + l2;
        const long long t0 = wall_clock64();
        if (lane == 0) ps_lds_add(c.ctl + PS_C_GATHER, 1u);
        for (;;) {
            const unsigned long long a = ps_peek(P.g_qkv, iq), b = ps_peek(P.g_qkv, ik), d = ps_peek(P.g_qkv, iv);
            const bool ok = (unsigned)(a >> 32) == ep && (unsigned)(b >> 32) == ep && (unsigned)(d >> 32) == ep;
            if (__all(ok)) {
                if (lane < HD / 2) {
                    reinterpret_cast<uint32_t *>(qs)[lane] = (uint32_t)a;
                    reinterpret_cast<uint32_t *>(kn)[lane] = (uint32_t)b;
                    reinterpret_cast<uint32_t *>(vn)[lane] = (uint32_t)d;
                }
                break;
            }
            if (ps_expired(c, t0, (0x20u << 8) | PS_ERR_GATHER)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) ps_lds_add(c.ctl + PS_C_GATHER, 0xFFFFFFFFu);
    }
    ps_cbar(c, 0x21);
    ps_stamp(c, 4);
    if (P.rope_order != 0) {
        if (tid < HD) {
            const int cc = tid < HD / 2 ? tid : tid - HD / 2;
            rope_apply(tid < HD / 2 ? qs : kn, cc, rope_cs, rope_sn, P.rope_order, P.rope_cols);
        }
        ps_cbar(c, 0x22);
    }
    // KV store of the new row (LayerKVCache::SetKRows / SetVRows, kv_cache.cc:159-249)
    if constexpr (Q8) {
        constexpr int NB = HD / 32;
        for (int b = w; b < 2 * NB; b += PS_NC) {
            half_t *src = b < NB ? kn : vn;
            const int bb = b < NB ? b : b - NB;
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
                    uint8_t *cache = b < NB ? ly.kcache : ly.vcache;
                    uint8_t *blk = cache + (size_t)pos * row_bytes + head_off + (size_t)bb * 34;
                    blk[2 + lane] = (uint8_t)(int8_t)qv;
                    if (lane == 0) *reinterpret_cast<uint16_t *>(blk) = __builtin_bit_cast(uint16_t, sch);
                }
                src[bb * 32 + lane] = f2h((float)qv * h2f(sch));
            }
        }
        ps_cbar(c, 0x23);
    } else {
        if (writer && tid < HD) {
            reinterpret_cast<half_t *>(ly.kcache + (size_t)pos * row_bytes + head_off)[tid] = kn[tid];
            reinterpret_cast<half_t *>(ly.vcache + (size_t)pos * row_bytes + head_off)[tid] = vn[tid];
        }
    }
    // scores
    const float alpha = 1.0f / sqrtf((float)HD) / P.kq_scale;
    const float mk = P.alibi ? alibi_slope(h + P.alibi_base, P.alibi_total) : 0.0f;
    for (int vw = w; vw < 4; vw += PS_NC) {
        float lmax = -INFINITY;
        for (int j = vw * 64 + lane; j < n_ctx; j += 256) {
            float cacc = 0.0f;
            const bool pre = vw == w && j < 256;         // this lane's prefetched row
            if (Q8 && j == pos) {
#pragma unroll 8
                for (int d = 0; d < HD; d++) cacc = __builtin_fmaf(h2f(qs[d]), h2f(kn[d]), cacc);
            } else {
                if constexpr (!Q8) {
                    if (j == pos) {
#pragma unroll
                        for (int i = 0; i < HD / 8; i++) {
                            const u32x4 t = reinterpret_cast<const u32x4 *>(kn)[i];
                            kreg[4 * i] = t[0]; kreg[4 * i + 1] = t[1]; kreg[4 * i + 2] = t[2]; kreg[4 * i + 3] = t[3];
                        }
                    } else if (!pre) load_k(j);
                } else if (!pre) load_k(j);
                if constexpr (Q8) {
#pragma unroll
                    for (int b = 0; b < HD / 32; b++) {
                        const float sc = hbits2f((uint16_t)(kbyte(b * 34) | (kbyte(b * 34 + 1) << 8)));
#pragma unroll
                        for (int i = 0; i < 32; i++) {
                            const int qv = (int)(int8_t)kbyte(b * 34 + 2 + i);
                            const float kvv = h2f(f2h((float)qv * sc));
                            cacc = __builtin_fmaf(h2f(qs[b * 32 + i]), kvv, cacc);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < HD / 2; i++) {
                        const half2_t k2 = __builtin_bit_cast(half2_t, kreg[i]);
                        cacc = __builtin_fmaf(h2f(qs[2 * i]), (float)k2[0], cacc);
                        cacc = __builtin_fmaf(h2f(qs[2 * i + 1]), (float)k2[1], cacc);
                    }
                }
            }
            half_t s = f2h(alpha * cacc);
            if (P.alibi) { float a = (float)j * mk; s = f2h(a + h2f(s)); }
            S[j] = s;
            lmax = fmaxf(lmax, P.kq_scale * h2f(s));
        }
        lmax = wave_max(lmax);
        if (lane == 0) red[vw] = lmax;
    }
    ps_cbar(c, 0x24);
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int vw = w; vw < 4; vw += PS_NC) {
        float lsum = 0.0f;
        for (int j = vw * 64 + lane; j < n_ctx; j += 256) {
            const float e = expf(P.kq_scale * h2f(S[j]) - mx);
            lsum += e;
            S[j] = f2h(e);
        }
        lsum = wave_sum(lsum);
        if (lane == 0) red[4 + vw] = lsum;
    }
    ps_cbar(c, 0x25);
    const float inv = 1.0f / (((red[4] + red[5]) + red[6]) + red[7]);
    for (int vw = w; vw < 4; vw += PS_NC)
        for (int j = vw * 64 + lane; j < n_ctx; j += 256) S[j] = f2h(h2f(S[j]) * inv);
    ps_cbar(c, 0x26);
    // O = P.V
    for (int vt = tid; vt < 256; vt += 64 * PS_NC) {
        const int dg = vt % DG, sp = vt / DG;
        const bool first = vt == tid;
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
        auto acc_q8 = [&](float pj, int j) {
            const uint8_t *blk = pvc + (size_t)j * row_bytes + head_off + (size_t)(dg / 4) * 34;
            const float sc = hbits2f(*reinterpret_cast<const uint16_t *>(blk));
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int qv = (int)(int8_t)blk[2 + (dg % 4) * 8 + e];
                o[e] = __builtin_fmaf(pj, h2f(f2h((float)qv * sc)), o[e]);
            }
        };
#pragma unroll
        for (int i = 0; i < VPRE; i++) {
            const int j = sp + NSPLIT * i;
            if (j < n_ctx) {
                const float pj = h2f(S[j]);
                if (j == pos) acc_new(pj);
                else if constexpr (Q8) {
                    if (first) {
                        const float sc = hbits2f(vq[i][0]);
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const int qv = (int)(int8_t)((vq[i][1 + (e >> 1)] >> (8 * (e & 1))) & 0xFF);
                            o[e] = __builtin_fmaf(pj, h2f(f2h((float)qv * sc)), o[e]);
                        }
                    } else acc_q8(pj, j);
                } else {
                    if (first) acc_v(pj, vreg[i]);
                    else acc_v(pj, reinterpret_cast<const u32x4 *>(pvc + (size_t)j * row_bytes + head_off)[dg]);
                }
            }
        }
        for (int j = sp + NSPLIT * VPRE; j < n_ctx; j += NSPLIT) {
            const float pj = h2f(S[j]);
            if (j == pos) acc_new(pj);
            else if constexpr (Q8) acc_q8(pj, j);
            else acc_v(pj, reinterpret_cast<const u32x4 *>(pvc + (size_t)j * row_bytes + head_off)[dg]);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) opart[sp * HD + dg * 8 + e] = o[e];
    }
    ps_cbar(c, 0x27);
    // combine in residue order, round to half, quantise the head's HD/32 blocks (dec_attn_emit_q8) and publish the image
    if (tid < HD) {
        float acc = opart[tid];
        for (int s2 = 1; s2 < NSPLIT; s2++) acc = acc + opart[s2 * HD + tid];
        const half_t yh = f2h(acc);
        if (P.dbg_att) P.dbg_att[(size_t)h * HD + tid] = yh;
        const float val = h2f(yh);
        const float bmx = half_wave_max(fabsf(val));
        const float qsc = bmx / 127;
        const int qv = q8_round_div1(val, qsc);
        const int sum = half_wave_sum_i32(qv);
        cod[tid] = (int8_t)qv;
        const unsigned ep = ps_epoch(layer, PS_E_ATT);
        if ((tid & 31) == 0) {
            const int blk = (h * HD + tid) >> 5;
            ps_publish(P.g_att, q_rows / 4 + blk, ep, __builtin_bit_cast(uint32_t, h2f(f2h(qsc))));
            ps_publish(P.g_att, q_rows / 4 + q_rows / 32 + blk, ep, __builtin_bit_cast(uint32_t, (float)sum));
        }
        PS_CB();
        if ((lane & 3) == 0) ps_publish(P.g_att, (h * HD + tid) >> 2, ep, *reinterpret_cast<const uint32_t *>(cod + tid));
    }
    ps_stamp(c, 5);
}

// ------------------------------------------------------------------ the loader wave
// The loader's own looks at the control words are inline asm: hipcc orders every LDS access it can see behind ALL
// outstanding direct-to-LDS loads (s_waitcnt vmcnt(0)), which would drain the stream at every look.
typedef __attribute__((address_space(3))) char ps_lds_char;
__device__ __forceinline__ unsigned ps_asm_lds_ld(unsigned addr)
{
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void ps_asm_lds_st(unsigned addr, unsigned v)
{
    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory");
}

struct PsLoad {
    uint32_t G, pub, free_to;
    unsigned ctl_addr;           // LDS byte address of the control words
};

// one op of one layer: S bytes starting at `base` into the ring, 1 KiB per instruction
__device__ __forceinline__ void ps_load_op(PsCtx &c, const uint8_t *base, uint32_t S, PsLoad &st, unsigned code)
{
    if (S == 0u) return;
    const int ninstr = (int)((S + 1023u) >> 10);
    constexpr int GRP = 4;      // instructions between two looks at the control words (one look costs an LDS round trip)
    for (int i0 = 0; i0 < ninstr; i0 += GRP) {
        const int ni = min(GRP, ninstr - i0);
        // ring space: the KiB about to be written must be behind every consumer wave's read position
        if ((int)(st.G + (uint32_t)ni * 1024u - st.free_to) > 0) {
            long long t0 = 0; bool timed = false;
            for (;;) {
                const unsigned n0 = ps_asm_lds_ld(st.ctl_addr + 4 * PS_C_NEED), n1 = ps_asm_lds_ld(st.ctl_addr + 4 * (PS_C_NEED + 1)),
                               n2 = ps_asm_lds_ld(st.ctl_addr + 4 * (PS_C_NEED + 2));
                unsigned mn = (int)(n1 - n0) < 0 ? n1 : n0;
                mn = (int)(n2 - mn) < 0 ? n2 : mn;
                st.free_to = mn + PS_RING;
                if ((int)(st.G + (uint32_t)ni * 1024u - st.free_to) <= 0) break;
                if (!timed) {
                    // blocked: everything issued so far may as well land and be published before sleeping
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if ((int)(st.G - st.pub) > 0) { st.pub = st.G; ps_asm_lds_st(st.ctl_addr + 4 * PS_C_FILLED, st.pub); }
                    t0 = wall_clock64(); timed = true;
                }
                __builtin_amdgcn_s_sleep(2);
                if (ps_expired(c, t0, (code << 8) | PS_ERR_SPACE)) break;
            }
        }
#pragma unroll
        for (int q = 0; q < GRP; q++) {
            if (q < ni) {
                uint32_t s = (uint32_t)(i0 + q) * 1024u + (uint32_t)c.lane * 16u;
                s = s > S - 16u ? S - 16u : s;       // the padding of the last KiB re-reads the op's last piece
                __builtin_amdgcn_global_load_lds((ps_glb_t *)(base + s), (ps_lds_t *)(c.smem + (st.G % PS_RING)), 16, 0, IFA_PS_NT ? 2 : 0);
                st.G += 1024u;
            }
        }
        // bounded depth; whatever is older than the depth has landed (loads return in order)
        const unsigned thin = ps_asm_lds_ld(st.ctl_addr + 4 * PS_C_GATHER);
        uint32_t landed;
        if (thin != 0u) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(IFA_PS_THIN - GRP) : "memory");
            landed = st.G - (uint32_t)(IFA_PS_THIN - GRP) * 1024u;
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(IFA_PS_INFLIGHT - GRP) : "memory");
            landed = st.G - (uint32_t)(IFA_PS_INFLIGHT - GRP) * 1024u;
        }
        if ((int)(landed - st.pub) > 0) { st.pub = landed; ps_asm_lds_st(st.ctl_addr + 4 * PS_C_FILLED, st.pub); }
    }
}

// ------------------------------------------------------------------ the kernel
// DT: weight format of all seven matrices; NJA / NJB: blocks per lane of a dim-wide / ffn-wide row; HD: head size;
// KVQ8: Q8_B32T2 KV cache.  Grid = one workgroup per CU (all must be resident: the host launches exactly the CU count).
template <int DT, int NJA, int NJB, int HD, bool KVQ8>
__global__ void __launch_bounds__(PS_THREADS) k_dec_persist(const PsParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PsCtx c;
    c.smem = smem;
    c.ctl = reinterpret_cast<unsigned *>(smem + PS_CTL_OFF);
    c.lane = threadIdx.x & 63;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.cu = blockIdx.x; c.ncu = gridDim.x;
    c.bar_gen = 0; c.filled_seen = 0; c.timeout = P.timeout_ticks;
    c.err = P.err; c.trace = nullptr; c.cur_layer = P.layer_begin;
    // control words: zeroed by the loader wave's lanes before anyone polls them -- every consumer's first wait is the
    // consumer barrier below, which itself needs the words zeroed: use a real workgroup barrier ONCE, before roles split
    if (threadIdx.x < PS_C_WORDS) c.ctl[threadIdx.x] = 0u;
    __syncthreads();

    const int q_rows = P.heads * HD, kv_rows = P.kv_heads * HD;
    const PsGeom g = ps_geom(P.dim, P.ffn, q_rows, kv_rows, P.row_bytes_a, P.row_bytes_b, c.ncu, c.cu);
    const int cap = block_capacity(DT);

    if (c.w == PS_NC) {
        // ---------------- loader
        PsLoad st; st.G = 0; st.pub = 0; st.free_to = PS_RING;
        st.ctl_addr = (unsigned)(uintptr_t)(ps_lds_char *)(smem + PS_CTL_OFF);
        for (int L = P.layer_begin; L < P.layer_end; L++) {
            c.cur_layer = L;
            if (ps_aborted(c)) break;
            const PsLayer &ly = P.layers[L];
            ps_load_op(c, ly.wqkv + (size_t)g.first[0] * P.row_bytes_a, (uint32_t)g.n[0] * P.row_bytes_a, st, 0x40);
            ps_load_op(c, ly.wo + (size_t)g.first[1] * P.row_bytes_a, (uint32_t)g.n[1] * P.row_bytes_a, st, 0x41);
            ps_load_op(c, ly.w13 + (size_t)g.first[2] * 2 * P.row_bytes_a, (uint32_t)g.n[2] * 2 * P.row_bytes_a, st, 0x42);
            ps_load_op(c, ly.w2 + (size_t)g.first[3] * P.row_bytes_b, (uint32_t)g.n[3] * P.row_bytes_b, st, 0x43);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ps_asm_lds_st(st.ctl_addr + 4 * PS_C_FILLED, st.G);
        return;
    }

    // ---------------- consumers
    const int maxcols = P.dim > P.ffn ? P.dim : P.ffn;
    char *img_base = smem + PS_IMG_OFF;
    char *stage_b = img_base + ps_img_bytes(maxcols);
    half_t *stage = reinterpret_cast<half_t *>(stage_b);
    half_t *res_o = reinterpret_cast<half_t *>(smem + PS_RES_OFF), *res_2 = res_o + PS_RES;
    const int pos = P.state[1];
    const int lane = c.lane, w = c.w;
    using FmtA = DecFmt<DT, NJA>;
    using FmtB = DecFmt<DT, NJB>;
    constexpr int MAXG_A = (NJA * 64 * block_capacity(DT) / 8 / 64 + PS_NC - 1) / PS_NC;      // 64-chunk groups per wave, dim-wide vector
    constexpr int MAXG_B = (NJB * 64 * block_capacity(DT) / 8 / 64 + PS_NC - 1) / PS_NC;
    constexpr int MAXL_A = (NJA * 64 * block_capacity(DT) / 2 / PS_NC + 63) / 64 + 1;           // granules per lane of a wave's third
    constexpr int MAXL_B = (NJB * 64 * block_capacity(DT) / 2 / PS_NC + 63) / 64 + 1;
    // head this CU serves (or -1)
    int my_head = -1;
    {
        const int stride = c.ncu / P.heads;
        const int hc = c.cu / stride;
        if (hc < P.heads && ps_head_cu(hc, P.heads, c.ncu) == c.cu) my_head = hc;
    }
    uint32_t lpos = 0;      // stream position of the current layer's first byte

    for (int L = P.layer_begin; L < P.layer_end; L++, lpos += g.layer_bytes) {
        c.cur_layer = L;
        if (ps_aborted(c)) break;
        const PsLayer &ly = P.layers[L];
        c.trace = (P.trace && L == P.trace_layer && w == 0) ? P.trace + (size_t)c.cu * 32 : nullptr;
        ps_stamp(c, 0);
        // ================= A: x -> RMSNorm -> Q8 -> wq | wk | wv rows
        {
            uint32_t *dst = reinterpret_cast<uint32_t *>(stage);
            const int ng = P.dim / 2;
            int f3, n3; ps_part(ng, PS_NC, w, f3, n3);
            if (L == P.layer_begin) {
                for (int i = f3 + lane; i < f3 + n3; i += 64) dst[i] = reinterpret_cast<const uint32_t *>(P.x_in)[i];
            } else {
                ps_gather<MAXL_A>(c, P.g_x, f3, n3, ps_epoch(L, PS_E_X), dst, 0x10);
            }
            ps_cbar(c, 0x11);
            ps_stamp(c, 1);
            // the residual values of this CU's wo rows (read by the Wo epilogue after `stage` has been reused)
            if (w == 0) for (int i = lane; i < g.n[1]; i += 64) res_o[i] = stage[g.first[1] + i];
            const PsImg img = ps_img(img_base, P.dim);
            ps_quantize<true, MAXG_A>(c, stage, P.dim, ly.attn_norm, ly.attn_norm_b, P.attn_norm_base, P.eps, img, 0x12);
            ps_stamp(c, 2);
            typename FmtA::X X;
            X.load(img.codes, img.scale, img.xsum, lane, P.nblk_a);
            const unsigned ep = ps_epoch(L, PS_E_QKV);
            ps_gemv<DT, NJA, 1>(c, X, P.nblk_a, P.row_bytes_a, lpos + g.off[0], lpos + g.off[1], g.n[0], 0x13,
                [&](int k, float a0, float) {
                    const int v = g.first[0] + k * PS_RB + min(lane, PS_RB - 1);
                    const bool in1 = v >= q_rows, in2 = v >= q_rows + kv_rows;
                    const half_t *bias = in2 ? ly.bv : (in1 ? ly.bk : ly.bq);
                    const int row = in2 ? v - q_rows - kv_rows : (in1 ? v - q_rows : v);
                    const half_t y = dec_bias(a0, bias, row);
                    ps_publish_pair(c, P.g_qkv, (g.first[0] + k * PS_RB) >> 1, ep, y);
                });
            ps_stamp(c, 3);
        }
        // ================= B: attention of this CU's head
        if (my_head >= 0) ps_attention<HD, KVQ8>(c, P, ly, L, my_head, pos, stage_b);
        // ================= C: quantised attention output -> wo rows (+ bias, + residual)
        {
            const int ng = q_rows / 4 + q_rows / 16;
            int f3, n3; ps_part(ng, PS_NC, w, f3, n3);
            ps_gather<MAXL_A>(c, P.g_att, f3, n3, ps_epoch(L, PS_E_ATT), reinterpret_cast<uint32_t *>(img_base), 0x30);
            ps_cbar(c, 0x31);
            ps_stamp(c, 6);
            const PsImg img = ps_img(img_base, q_rows);
            typename FmtA::X X;
            X.load(img.codes, img.scale, img.xsum, lane, P.nblk_a);
            const unsigned ep = ps_epoch(L, PS_E_A);
            ps_gemv<DT, NJA, 1>(c, X, P.nblk_a, P.row_bytes_a, lpos + g.off[1], lpos + g.off[2], g.n[1], 0x32,
                [&](int k, float a0, float) {
                    const int i = k * PS_RB + min(lane, PS_RB - 1);
                    half_t y = dec_bias(a0, ly.bo, g.first[1] + i);
                    y = f2h(h2f(res_o[i]) + h2f(y));
                    ps_publish_pair(c, P.g_a, (g.first[1] + k * PS_RB) >> 1, ep, y);
                });
            ps_stamp(c, 7);
        }
        // ================= D: a -> RMSNorm -> Q8 -> w1, w3 rows -> act(t1) * t2
        {
            uint32_t *dst = reinterpret_cast<uint32_t *>(stage);
            int f3, n3; ps_part(P.dim / 2, PS_NC, w, f3, n3);
            ps_gather<MAXL_A>(c, P.g_a, f3, n3, ps_epoch(L, PS_E_A), dst, 0x50);
            ps_cbar(c, 0x51);
            ps_stamp(c, 8);
            if (w == 0) for (int i = lane; i < g.n[3]; i += 64) res_2[i] = stage[g.first[3] + i];
            const PsImg img = ps_img(img_base, P.dim);
            ps_quantize<true, MAXG_A>(c, stage, P.dim, ly.ffn_norm, ly.ffn_norm_b, P.ffn_norm_base, P.eps, img, 0x52);
            ps_stamp(c, 9);
            typename FmtA::X X;
            X.load(img.codes, img.scale, img.xsum, lane, P.nblk_a);
            const unsigned ep = ps_epoch(L, PS_E_ACT);
            ps_gemv<DT, NJA, 2>(c, X, P.nblk_a, P.row_bytes_a, lpos + g.off[2], lpos + g.off[3], g.n[2] * 2, 0x53,
                [&](int k, float a0, float a1) {
                    const int row = g.first[2] + k * PS_RB + min(lane, PS_RB - 1);
                    half_t y = dec_bias(a0, ly.b1, row);
                    const half_t t2 = dec_bias(a1, ly.b3, row);
                    const half_t act = f2h(act_fn(h2f(y), P.act_kind));
                    y = f2h(h2f(act) * h2f(t2));
                    ps_publish_pair(c, P.g_act, (g.first[2] + k * PS_RB) >> 1, ep, y);
                });
            ps_stamp(c, 10);
        }
        // ================= E: gated product -> Q8 -> w2 rows (+ bias, + residual) -> next layer's x
        {
            uint32_t *dst = reinterpret_cast<uint32_t *>(stage);
            int f3, n3; ps_part(P.ffn / 2, PS_NC, w, f3, n3);
            ps_gather<MAXL_B>(c, P.g_act, f3, n3, ps_epoch(L, PS_E_ACT), dst, 0x60);
            ps_cbar(c, 0x61);
            ps_stamp(c, 11);
            const PsImg img = ps_img(img_base, P.ffn);
            ps_quantize<false, MAXG_B>(c, stage, P.ffn, nullptr, nullptr, 0.0f, P.eps, img, 0x62);
            ps_stamp(c, 12);
            typename FmtB::X X;
            X.load(img.codes, img.scale, img.xsum, lane, P.nblk_b);
            const bool last = L + 1 == P.layer_end;
            const unsigned ep = ps_epoch(L + 1, PS_E_X);
            ps_gemv<DT, NJB, 1>(c, X, P.nblk_b, P.row_bytes_b, lpos + g.off[3], lpos + g.layer_bytes, g.n[3], 0x63,
                [&](int k, float a0, float) {
                    const int i = k * PS_RB + min(lane, PS_RB - 1);
                    half_t y = dec_bias(a0, ly.b2, g.first[3] + i);
                    y = f2h(h2f(res_2[i]) + h2f(y));
                    if (last) { if (lane < PS_RB) P.x_out[g.first[3] + i] = y; }
                    else ps_publish_pair(c, P.g_x, (g.first[3] + k * PS_RB) >> 1, ep, y);
                });
            ps_stamp(c, 13);
        }
    }
}

} // namespace ifa
