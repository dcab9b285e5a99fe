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
#include "ifa_decode_attn.h"

namespace ifa {

typedef __attribute__((address_space(1))) unsigned long long ps_gu64;
typedef __attribute__((address_space(1))) unsigned int ps_gu32;
typedef __attribute__((address_space(3))) void ps_lds_t;
typedef const __attribute__((address_space(1))) void ps_glb_t;
// a pointer the kernel got out of the layer table is "generic" to the compiler (flat_load: both wait counters, no
// saddr form): tell it the data is global
template <typename T> __device__ __forceinline__ const __attribute__((address_space(1))) T *ps_g(const T *p) { return (const __attribute__((address_space(1))) T *)p; }
template <typename T> __device__ __forceinline__ __attribute__((address_space(1))) T *ps_gw(T *p) { return (__attribute__((address_space(1))) T *)p; }

// consumer waves per workgroup.  One wave per SIMD issues an instruction every ~4-5 cycles whatever it is
// (MI355X_MICROARCH.md, "one wave per SIMD"): with 3 consumers the Q8 quantiser, the 4-bit decode and the epilogues were
// issue-bound (65 us per layer); 7 consumers + the loader = two waves per SIMD.
#ifndef IFA_PS_NC
#define IFA_PS_NC 6
#endif
constexpr int PS_NC = IFA_PS_NC;
// loader waves per workgroup: ONE wave's direct-to-LDS stream tops out at ~4.9 TB/s over the chip whatever its depth
// (32 or 60 KiB in flight, nt or not); two waves taking the 4 KiB groups in turn reach 6.4 TB/s, four 6.6 (measured with
// the consumers switched off, option persist_depth >= 8)
constexpr int PS_NL = 2;
constexpr int PS_THREADS = 64 * (PS_NC + PS_NL);
constexpr uint32_t PS_RING = 112u * 1024u;       // LDS ring of the weight stream (7 x 16 KiB)
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
enum { PS_C_FILLED = 0 /* PS_NL <= 4 words */, PS_C_NEED = 4 /* PS_NC <= 15 words */, PS_C_BAR = 20, PS_C_ABORT = 21, PS_C_GATHER = 22, PS_C_PART = 32 /* 48 floats */, PS_C_WORDS = 128 };
static_assert(PS_NL >= 1 && PS_NL <= 4, "loader waves");
static_assert(PS_RING % 4096u == 0, "the loader writes the ring in 4 KiB groups");
static_assert(PS_NC >= 4 && PS_NC <= 15, "consumer waves: the attention maps k_dec_attn's 256 threads onto waves 0-3");
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
    // biases in the same order as the streamed rows: bq | bk | bv in one vector, b1 / b3 interleaved (null = none).  One
    // pointer per op on purpose: a per-lane choice between several pointers makes hipcc index the variables that hold
    // them, which puts them -- and everything captured next to them -- into scratch memory
    const half_t *bqkv, *bo, *b13, *b2;
    uint8_t *kcache, *vcache;
};

// The table is read through the constant address space: a uniform-index load from it is a SCALAR load.  As plain global
// memory hipcc fetched every pointer with a vector load and waited vmcnt(0) for it -- in the consumer waves that wait
// also covers the write-through granule stores issued just before (~2 us each until acknowledged), in the loader wave
// it drains the weight stream at every op.
typedef const __attribute__((address_space(4))) PsLayer ps_layer_c;

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
    int tune_depth, tune_prio;   // tuning knobs of the loader (options persist_depth / persist_prio)
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
        o += (bytes + 4095u) & ~4095u;
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

// (the slow path of the waits is a real function: inlined into every spin loop it was a good part of a kernel whose
// layer loop has to stay inside the instruction cache.  It takes VALUES only: a reference to the context would put the
// context into scratch memory, and every later use of it would be a vector memory load.)
// returns the (possibly initialised) start time, or -1 to stop waiting
__device__ __attribute__((noinline)) long long ps_expired_slow(unsigned *ctl, unsigned *err, unsigned timeout, long long t0, unsigned code, int lane,
                                                                int cu, int layer, int w)
{
    if (__hip_atomic_load(ctl + PS_C_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return -1;
    const long long now = wall_clock64();
    if (t0 == 0) return now;
    if (now - t0 > (long long)timeout) {
        __hip_atomic_store(ctl + PS_C_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0) {
            if (atomicCAS(err, 0u, code) == 0u) { err[1] = (unsigned)cu; err[2] = (unsigned)layer; err[3] = (unsigned)w; }
        }
        return -1;
    }
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {      // someone else failed
        __hip_atomic_store(ctl + PS_C_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return -1;
    }
    return t0;
}

// Inside a spin loop: true = stop waiting (this workgroup or another one gave up).  The clock (s_memrealtime: a memory
// instruction) and the abort words are looked at every 32nd spin only -- a wait that is satisfied at once never touches
// them.
struct PsSpin { unsigned n = 0; long long t0 = 0; };
__device__ __forceinline__ bool ps_expired(PsCtx &c, PsSpin &sp, unsigned code)
{
    if ((++sp.n & 31u) != 0u) return false;
    const long long r = ps_expired_slow(c.ctl, c.err, c.timeout, sp.t0, code, c.lane, c.cu, c.cur_layer, c.w);
    if (r < 0) return true;
    sp.t0 = r;
    return false;
}
__device__ __forceinline__ bool ps_aborted(const PsCtx &c) { return ps_lds_ld(c.ctl + PS_C_ABORT) != 0u; }

// barrier over the consumer waves (the loader never joins a barrier: it would stop streaming)
__device__ __forceinline__ void ps_cbar(PsCtx &c, unsigned code)
{
    c.bar_gen += PS_NC;
    PS_CB();
    if (c.lane == 0) ps_lds_add(c.ctl + PS_C_BAR, 1u);
    PsSpin sp;
    while ((int)(ps_lds_ld(c.ctl + PS_C_BAR) - c.bar_gen) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (ps_expired(c, sp, (code << 8) | PS_ERR_BAR)) break;
    }
    PS_CB();
}

// wait until the loader has landed the stream up to byte position `end`
__device__ __forceinline__ void ps_wait_filled(PsCtx &c, uint32_t end, unsigned code)
{
    if ((int)(c.filled_seen - end) >= 0) return;
    PsSpin sp;
    for (;;) {
        // each loader wave publishes the stream position of its first group that has not landed yet
        unsigned f = ps_lds_ld(c.ctl + PS_C_FILLED);
#pragma unroll
        for (int q = 1; q < PS_NL; q++) {
            const unsigned fq = ps_lds_ld(c.ctl + PS_C_FILLED + q);
            f = (int)(fq - f) < 0 ? fq : f;
        }
        c.filled_seen = f;
        if ((int)(c.filled_seen - end) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
        if (ps_expired(c, sp, (code << 8) | PS_ERR_RING)) break;
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

// This wave collects granules [first, first + n) of `g`: every pass requests the 64-granule groups that are not complete
// yet (all of them the first time), a group whose 64 tags all equal `epoch` is written to dst[granule index] (LDS dwords)
// and not read again.  When the producers are done before the consumer arrives -- the common case -- that is ONE memory
// round trip; stragglers cost re-reads of their groups only.
template <int MAXL>
__device__ __forceinline__ void ps_gather(PsCtx &c, const unsigned long long *g, int first, int n, unsigned epoch, uint32_t *dst, unsigned code)
{
    if (n <= 0) return;
    if (c.lane == 0) ps_lds_add(c.ctl + PS_C_GATHER, 1u);
    PsSpin sp;
    for (int base = 0; base < n; base += 64 * MAXL) {
        unsigned long long done = 0ull;          // bit k: group k of this chunk is in LDS (wave-uniform)
        const int ngrp = min(MAXL, (n - base + 63) >> 6);
        for (;;) {
            unsigned long long x[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) {
                if (k < ngrp && !((done >> k) & 1ull)) x[k] = ps_peek(g, first + min(base + c.lane + 64 * k, n - 1));
            }
            bool all = true;
#pragma unroll
            for (int k = 0; k < MAXL; k++) {
                if (k < ngrp && !((done >> k) & 1ull)) {
                    if (__all((unsigned)(x[k] >> 32) == epoch)) {
                        const int idx = base + c.lane + 64 * k;
                        if (idx < n) dst[first + idx] = (uint32_t)x[k];
                        done |= 1ull << k;
                    } else all = false;
                }
            }
            if (all) break;
            if (ps_expired(c, sp, (code << 8) | PS_ERR_GATHER)) break;
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

// norm weights of this wave's 64-chunk groups: requested BEFORE the gather of the vector they scale (they do not depend on
// it), so that their HBM / L2 latency is not a serial step behind the hand-off
template <int MAXG>
struct PsNormW {
    half8_t wv[MAXG], bv[MAXG];
    __device__ __forceinline__ void request(const PsCtx &c, const half_t *nw, const half_t *nb, int cols)
    {
        const int chunks = cols >> 3;
#pragma unroll
        for (int gi = 0; gi < MAXG; gi++) {
            const int ch = 64 * (c.w + PS_NC * gi) + c.lane;
            if (ch < chunks) {
                if (nw) wv[gi] = *reinterpret_cast<const __attribute__((address_space(1))) half8_t *>(ps_g(nw) + (size_t)ch * 8);
                if (nb) bv[gi] = *reinterpret_cast<const __attribute__((address_space(1))) half8_t *>(ps_g(nb) + (size_t)ch * 8);
            }
        }
    }
};

// [RMS-normalise and] quantise the staged F16 vector into the image: the arithmetic of XPre::finish (same chunk -> lane
// mapping of the canonical RMS order, same quad-local Q8_B32T2 quantiser), the 64-chunk groups dealt to the consumer waves.
template <bool NORM, int MAXG>
__device__ __forceinline__ void ps_quantize(PsCtx &c, const half_t *stage, int cols, const PsNormW<NORM ? MAXG : 1> &NW, bool has_w, bool has_b,
                                            float multi_base, float eps, const PsImg &L, unsigned code)
{
    const int chunks = cols >> 3, ngroups = (chunks + 63) >> 6;
    float *part = reinterpret_cast<float *>(c.ctl + PS_C_PART);
    float scale = 1.0f;
    if constexpr (NORM) {
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
        if (code == 0x30u) ps_stamp(c, 14);
        ps_cbar(c, code);
        if (code == 0x30u) ps_stamp(c, 15);
        scale = rms_scale_of(rms_total(part, ngroups), cols, eps);
    }
    constexpr int UNR = NORM ? MAXG : 1;       // the norm weights live in registers: static indices only
#pragma unroll UNR
    for (int gi = 0; gi < MAXG; gi++) {
        const int ch = 64 * (c.w + PS_NC * gi) + c.lane;
        if (ch >= chunks) continue;          // whole quads (4 lanes = one block) are in or out together
        const half8_t xv = *reinterpret_cast<const half8_t *>(stage + (size_t)ch * 8);
        float v[8];
        if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float t = (float)xv[i] * scale;
                if (has_w) {
                    float m = multi_base + (float)NW.wv[gi][i];
                    t = t * m;
                    if (has_b) t = t + (float)NW.bv[gi][i];
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
    if (code == 0x30u) ps_stamp(c, 16);
    ps_cbar(c, code);
}

// Unit rows of one op out of the ring (a unit = one tiled weight row; the gated FFN's units alternate w1 / w3 rows).  A
// batch = `ub` consecutive units of this CU's stream segment (ub <= UMAX); batches are dealt to the consumer waves
// round-robin, the last one may be short.  The batch loop is a real loop and UMAX is small: the layer loop's code has to
// stay inside the instruction cache (a fully unrolled first version was 150 KB and ran at ~17 cycles per instruction).
// Same lane -> block mapping, dot() and wave_sum() as k_dec_gemv; epi(u0, nu, a) is called by every lane, lane i < nu
// holding the sum of unit u0 + i.
template <int DT, int NJ, int UMAX, class Epi>
__device__ __forceinline__ void ps_gemv(PsCtx &c, const typename DecFmt<DT, NJ>::X &X, int nblk, uint32_t row_bytes, uint32_t op_pos,
                                        uint32_t op_end, int n_units, int ub, unsigned code, Epi &&epi)
{
    using Fmt = DecFmt<DT, NJ>;
    const int nb = (n_units + ub - 1) / ub;
#pragma nounroll
    for (int k = c.w; k < nb; k += PS_NC) {
        const int u0 = k * ub, nu = min(ub, n_units - u0);
        const uint32_t pos0 = op_pos + (uint32_t)u0 * row_bytes;
        if (code == 0x42u && k < 3 * PS_NC) ps_stamp(c, 17 + 3 * (k / PS_NC));
        ps_wait_filled(c, pos0 + (uint32_t)nu * row_bytes, code);
        if (code == 0x42u && k < 3 * PS_NC) ps_stamp(c, 18 + 3 * (k / PS_NC));
        typename Fmt::W wr[UMAX];
#pragma unroll
        for (int i = 0; i < UMAX; i++) {
            // units past the batch re-read its first one (defined registers, result unused)
            const WSrcLdsRing<PS_RING> src = {c.smem, (pos0 + (uint32_t)(i < nu ? i : 0) * row_bytes) % PS_RING};
            wr[i].load_src(src, nblk, c.lane);
        }
        // LDS executes a wave's instructions in order: this store is behind the reads above, so the loader sees the
        // space as free only after they were served
        PS_CB();
        if (c.lane == 0) ps_lds_st(c.ctl + PS_C_NEED + c.w, k + PS_NC < nb ? op_pos + (uint32_t)(k + PS_NC) * (uint32_t)ub * row_bytes : op_end);
        PS_CB();
        float a[UMAX];
#pragma unroll
        for (int i = 0; i < UMAX; i++) a[i] = wr[i].dot(X);
#pragma unroll
        for (int i = 0; i < UMAX; i++) a[i] = wave_sum(a[i]);
        float mine = 0.0f;
#pragma unroll
        for (int i = 0; i < UMAX; i++) {
            if (c.lane == i) mine = a[i];
        }
        epi(u0, nu, mine);
        if (code == 0x42u && k < 3 * PS_NC) ps_stamp(c, 19 + 3 * (k / PS_NC));
    }
    if (nb <= c.w && c.lane == 0) ps_lds_st(c.ctl + PS_C_NEED + c.w, op_end);
}

// units per batch: one batch per wave when the CU's share is small, UMAX otherwise; a multiple of `quantum` (2 rows = one
// granule of output; 4 units = one granule of the gated product)
__device__ __forceinline__ int ps_units_per_batch(int n_units, int umax, int quantum)
{
    const int per_wave = ((n_units + PS_NC - 1) / PS_NC + quantum - 1) / quantum * quantum;
    return max(quantum, min(umax, per_wave));
}

// ------------------------------------------------------------------ attention of one head on this CU
// The arithmetic and its ORDER are k_dec_attn's (scores: fp32 fma chain in d order; S and P rounded to half; maximum and
// sum per wave of 64 keys, combined in wave order; P.V per (key residue, 8-dim group) thread, partials added in residue
// order): consumer waves 0-3 ARE that kernel's 256 threads, the other consumer waves only join the barriers.
template <int HD, bool Q8>
__device__ __forceinline__ void ps_attention(PsCtx &c, const PsParams &P, const ps_layer_c &ly, int layer, int h, int pos, char *scratch)
{
    // VPRE: V rows requested up front per thread (keys < NSPLIT * VPRE); 8 = 32 registers next to the 64 of the key row
    // (two waves per SIMD: 256 registers per lane), later keys are requested where they are used
    constexpr int DG = HD / 8, NSPLIT = 256 / DG, VPRE = (256 / NSPLIT) < 8 ? (256 / NSPLIT) : 8;
    static_assert(HD == 64 || HD == 128, "persistent attention: head_dim 64 or 128");
    half_t *qs = reinterpret_cast<half_t *>(scratch);
    half_t *kn = qs + HD, *vn = kn + HD;
    float *red = reinterpret_cast<float *>(vn + HD);                  // [16]
    int8_t *cod = reinterpret_cast<int8_t *>(red + 16);               // [HD]
    float *opart = reinterpret_cast<float *>(cod + HD);               // [NSPLIT][HD]
    half_t *S = reinterpret_cast<half_t *>(opart + NSPLIT * HD);      // [n_ctx]
    const int lane = c.lane, w = c.w, tid = w * 64 + lane;
    const bool act = w < 4;                                           // one of the 256 attention threads
    const int n_ctx = pos + 1;
    const int group = P.heads / P.kv_heads, kvh = h / group;
    const bool writer = (h % group) == 0;
    const int q_rows = P.heads * HD, kv_dim = P.kv_heads * HD;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const size_t head_off = Q8 ? (size_t)((kvh * HD) / 32) * 34 : (size_t)kvh * HD * 2;
    const uint8_t *pkc = ly.kcache, *pvc = ly.vcache;
    uint8_t *kcw = ly.kcache, *vcw = ly.vcache;

    constexpr int KBYTES = (HD / 32) * 34;
    constexpr int KALIGN = HD == 128 ? 8 : 4;
    uint32_t kreg[Q8 ? 1 : HD / 2];
    uint32_t kq32[Q8 ? KBYTES / 4 : 1];
    auto load_k = [&](int j) __attribute__((always_inline)) {
        const __attribute__((address_space(1))) uint8_t *rowp = ps_g(pkc) + (size_t)j * row_bytes + head_off;
        if constexpr (Q8) {
            if constexpr (KALIGN == 8) {
#pragma unroll
                for (int i = 0; i < KBYTES / 8; i++) {
                    const u32x2 t = reinterpret_cast<const __attribute__((address_space(1))) u32x2 *>(rowp)[i];
                    kq32[2 * i] = t[0]; kq32[2 * i + 1] = t[1];
                }
            } else {
#pragma unroll
                for (int i = 0; i < KBYTES / 4; i++) kq32[i] = reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(rowp)[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < HD / 8; i++) {
                const u32x4 t = reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(rowp)[i];
                kreg[4 * i] = t[0]; kreg[4 * i + 1] = t[1]; kreg[4 * i + 2] = t[2]; kreg[4 * i + 3] = t[3];
            }
        }
    };
    auto kbyte = [&](int B) __attribute__((always_inline)) -> uint32_t { return (kq32[B >> 2] >> (8 * (B & 3))) & 0xFFu; };

    // requests first: this thread's key row (the caches hold >= DEC_ATTN_MIN_ROWS rows: no clamp).  Its V rows are
    // requested after the scores, when the key row's 64 registers are free (three waves per SIMD: 168 registers a lane);
    // their latency then overlaps the softmax barriers.
    const int dg = tid % DG, sp = (tid / DG) % NSPLIT;
    u32x4 vreg[Q8 ? 1 : VPRE];
    uint16_t vq[Q8 ? VPRE : 1][5];
    float rope_cs = 1.0f, rope_sn = 0.0f;
    if (act) {
        load_k(tid);
        if (P.rope_order != 0) {
            const int cc = min(tid < HD / 2 ? tid : tid - HD / 2, HD / 2 - 1);
            rope_cs = P.rope_tab[2 * cc]; rope_sn = P.rope_tab[2 * cc + 1];
        }
    }
    // the new token's q | k | v of this head: three granule runs of HD/2 each, gathered by wave 0
    if (w == 0) {
        const unsigned ep = ps_epoch(layer, PS_E_QKV);
        const int l2 = min(lane, HD / 2 - 1);
        const int iq = (h * HD) / 2 + l2, ik = (q_rows + kvh * HD) / 2 + l2, iv = (q_rows + kv_dim + kvh * HD) / 2 + l2;
        PsSpin spn;
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
            if (ps_expired(c, spn, (0x50u << 8) | PS_ERR_GATHER)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) ps_lds_add(c.ctl + PS_C_GATHER, 0xFFFFFFFFu);
    }
    ps_cbar(c, 0x51);
    ps_stamp(c, 4);
    if (P.rope_order != 0) {
        if (tid < HD) {
            const int cc = tid < HD / 2 ? tid : tid - HD / 2;
            rope_apply(tid < HD / 2 ? qs : kn, cc, rope_cs, rope_sn, P.rope_order, P.rope_cols);
        }
        ps_cbar(c, 0x52);
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
                    uint8_t *cache = b < NB ? kcw : vcw;
                    uint8_t *blk = cache + (size_t)pos * row_bytes + head_off + (size_t)bb * 34;
                    blk[2 + lane] = (uint8_t)(int8_t)qv;
                    if (lane == 0) *reinterpret_cast<uint16_t *>(blk) = __builtin_bit_cast(uint16_t, sch);
                }
                src[bb * 32 + lane] = f2h((float)qv * h2f(sch));
            }
        }
        ps_cbar(c, 0x53);
    } else {
        if (writer && tid < HD) {
            reinterpret_cast<half_t *>(kcw + (size_t)pos * row_bytes + head_off)[tid] = kn[tid];
            reinterpret_cast<half_t *>(vcw + (size_t)pos * row_bytes + head_off)[tid] = vn[tid];
        }
    }
    // scores: one key per thread, fp32 fma in d order (Gemm_Alg2_Kernel order, products exact)
    const float alpha = 1.0f / sqrtf((float)HD) / P.kq_scale;
    const float mk = P.alibi ? alibi_slope(h + P.alibi_base, P.alibi_total) : 0.0f;
    float lmax = -INFINITY;
    if (act) {
        for (int j = tid; j < n_ctx; j += 256) {
            float cacc = 0.0f;
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
                    } else if (j >= 256) load_k(j);
                } else if (j >= 256) load_k(j);
                if constexpr (Q8) {
#pragma unroll
                    for (int b = 0; b < HD / 32; b++) {
                        const half_t sch = __builtin_bit_cast(half_t, (uint16_t)(kbyte(b * 34) | (kbyte(b * 34 + 1) << 8)));
                        const half2_t sc2 = {sch, sch};
#pragma unroll
                        for (int w4 = 0; w4 < 8; w4++) {      // four codes per dword (q8x4_dequant_h, ifa_decode_kernels.h)
                            const int B0 = b * 34 + 2 + 4 * w4;
                            const uint32_t cw = (B0 & 3) == 0 ? kq32[B0 >> 2]
                                : __builtin_amdgcn_alignbyte(kq32[(B0 >> 2) + 1 < KBYTES / 4 ? (B0 >> 2) + 1 : (B0 >> 2)], kq32[B0 >> 2], (B0 & 3));
                            half2_t lo, hi;
                            q8x4_dequant_h(cw, sc2, lo, hi);
                            cacc = __builtin_fmaf(h2f(qs[b * 32 + 4 * w4]), (float)lo[0], cacc);
                            cacc = __builtin_fmaf(h2f(qs[b * 32 + 4 * w4 + 1]), (float)lo[1], cacc);
                            cacc = __builtin_fmaf(h2f(qs[b * 32 + 4 * w4 + 2]), (float)hi[0], cacc);
                            cacc = __builtin_fmaf(h2f(qs[b * 32 + 4 * w4 + 3]), (float)hi[1], cacc);
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
            half_t sv = f2h(alpha * cacc);
            if (P.alibi) { float a = (float)j * mk; sv = f2h(a + h2f(sv)); }
            S[j] = sv;
            lmax = fmaxf(lmax, P.kq_scale * h2f(sv));
        }
        lmax = wave_max(lmax);
        if (lane == 0) red[w] = lmax;
#pragma unroll
        for (int i = 0; i < VPRE; i++) {
            const int j = sp + NSPLIT * i;
            if constexpr (!Q8) {
                vreg[i] = reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(ps_g(pvc) + (size_t)j * row_bytes + head_off)[dg];
            } else {
                const __attribute__((address_space(1))) uint16_t *blk =
                    reinterpret_cast<const __attribute__((address_space(1))) uint16_t *>(ps_g(pvc) + (size_t)j * row_bytes + head_off + (size_t)(dg / 4) * 34);
                vq[i][0] = blk[0];
#pragma unroll
                for (int e = 0; e < 4; e++) vq[i][1 + e] = blk[1 + (dg % 4) * 4 + e];
            }
        }
    }
    ps_cbar(c, 0x54);
    if (act) {
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float lsum = 0.0f;
        for (int j = tid; j < n_ctx; j += 256) {
            const float e = expf(P.kq_scale * h2f(S[j]) - mx);
            lsum += e;
            S[j] = f2h(e);
        }
        lsum = wave_sum(lsum);
        if (lane == 0) red[4 + w] = lsum;
    }
    ps_cbar(c, 0x55);
    if (act) {
        const float inv = 1.0f / (((red[4] + red[5]) + red[6]) + red[7]);
        for (int j = tid; j < n_ctx; j += 256) S[j] = f2h(h2f(S[j]) * inv);
    }
    ps_cbar(c, 0x56);
    // O = P.V : thread (sp, dg) accumulates keys j = sp + NSPLIT * i for its 8 dims
    if (act) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = 0.0f;
        auto acc_v = [&](float pj, const u32x4 vv) __attribute__((always_inline)) {
            const half8_t v8 = __builtin_bit_cast(half8_t, vv);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = __builtin_fmaf(pj, (float)v8[e], o[e]);
        };
        auto acc_new = [&](float pj) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = __builtin_fmaf(pj, h2f(vn[dg * 8 + e]), o[e]);
        };
        auto acc_q8w = [&](float pj, uint16_t scb, uint16_t c0, uint16_t c1, uint16_t c2, uint16_t c3) __attribute__((always_inline)) {
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
        auto acc_q8 = [&](float pj, int j) __attribute__((always_inline)) {
            const __attribute__((address_space(1))) uint16_t *blk =
                reinterpret_cast<const __attribute__((address_space(1))) uint16_t *>(ps_g(pvc) + (size_t)j * row_bytes + head_off + (size_t)(dg / 4) * 34);
            const __attribute__((address_space(1))) uint16_t *cp = blk + 1 + (dg % 4) * 4;
            acc_q8w(pj, blk[0], cp[0], cp[1], cp[2], cp[3]);
        };
#pragma unroll
        for (int i = 0; i < VPRE; i++) {        // first 256 keys: V rows already in registers (static indexing)
            const int j = sp + NSPLIT * i;
            if (j < n_ctx) {
                const float pj = h2f(S[j]);
                if (j == pos) acc_new(pj);
                else if constexpr (Q8) acc_q8w(pj, vq[i][0], vq[i][1], vq[i][2], vq[i][3], vq[i][4]);
                else acc_v(pj, vreg[i]);
            }
        }
        for (int j = sp + NSPLIT * VPRE; j < n_ctx; j += NSPLIT) {
            const float pj = h2f(S[j]);
            if (j == pos) acc_new(pj);
            else if constexpr (Q8) acc_q8(pj, j);
            else acc_v(pj, reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(ps_g(pvc) + (size_t)j * row_bytes + head_off)[dg]);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) opart[sp * HD + dg * 8 + e] = o[e];
    }
    ps_cbar(c, 0x57);
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
    return __builtin_amdgcn_readfirstlane(v);      // (every lane read the same word: keep what follows on the scalar unit)
}
__device__ __forceinline__ void ps_asm_lds_st(unsigned addr, unsigned v)
{
    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory");
}

struct PsLoad {
    uint32_t G, roff, pub, free_to;      // this wave's next 4 KiB group (stream position), G % PS_RING, published, ring space limit
    uint32_t op_pos;                     // stream position of the current op's first byte
    unsigned ctl_addr;                   // LDS byte address of the control words
    unsigned li;                         // which of the PS_NL loader waves
    unsigned issued;                     // groups this wave has issued
    unsigned blocked;                    // times the ring was full (tuning)
    bool free_run;                       // tuning: stream without consumers (ring space never checked)
    int variant;
};

// Two loader waves share a SIMD each with a consumer wave and get an issue slot every ~8-10 cycles; at 25 GB/s per CU a
// KiB is due every ~100 cycles, so the loop below is counted in instructions: groups of four 1-KiB loads off ONE
// per-lane offset (immediate offsets 0 / 1 / 2 / 3 KiB), the control words looked at once per group.  An op's stream is
// padded to 4 KiB in the ring (the padding is never read), so the whole launch is one sequence of 4 KiB groups: group n
// belongs to loader wave n % PS_NL, never straddles an op boundary or the end of the ring (112 KiB = 28 groups).
__device__ __forceinline__ bool ps_ring_space(PsCtx &c, PsLoad &st, unsigned code)
{
    PsSpin sp; bool timed = false;
    for (;;) {
        unsigned mn = ps_asm_lds_ld(st.ctl_addr + 4 * PS_C_NEED);
#pragma unroll
        for (int q = 1; q < PS_NC; q++) {
            const unsigned nq = ps_asm_lds_ld(st.ctl_addr + 4 * (PS_C_NEED + q));
            mn = (int)(nq - mn) < 0 ? nq : mn;
        }
        st.free_to = mn + PS_RING;
        if ((int)(st.G + 4096u - st.free_to) <= 0) return true;
        if (!timed) {
            st.blocked++;
            // blocked: everything issued so far may as well land and be published before sleeping
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((int)(st.G - st.pub) > 0) { st.pub = st.G; ps_asm_lds_st(st.ctl_addr + 4 * (PS_C_FILLED + st.li), st.pub); }
            timed = true;
        }
        __builtin_amdgcn_s_sleep(2);
        if (ps_expired(c, sp, (code << 8) | PS_ERR_SPACE)) return false;
    }
}

// one op of one layer: S bytes starting at `base`, at stream position st.op_pos
template <int NTV, bool DEEP>
__device__ __forceinline__ void ps_load_op_t(PsCtx &c, const uint8_t *base, uint32_t S, PsLoad &st, unsigned code)
{
    const uint32_t op_end = st.op_pos + ((S + 4095u) & ~4095u);
    const uint32_t last = S - 16u;
    constexpr uint32_t STEP = 4096u * PS_NL;
    while ((int)(st.G - op_end) < 0) {
        const uint32_t goff = st.G - st.op_pos;              // this group's offset in the op
        const uint32_t voff = goff + (uint32_t)c.lane * 16u;
        // ring space: the 4 KiB about to be written must be behind every consumer wave's read position
        if (!st.free_run && (int)(st.G + 4096u - st.free_to) > 0) { if (!ps_ring_space(c, st, code)) break; }
        char *dst = c.smem + st.roff;
        if (goff + 4096u <= S) {
            // (the instruction's immediate offset moves BOTH the global address and the LDS address: M0 stays put)
            __builtin_amdgcn_global_load_lds((ps_glb_t *)(base + voff), (ps_lds_t *)dst, 16, 0, NTV);
            __builtin_amdgcn_global_load_lds((ps_glb_t *)(base + voff), (ps_lds_t *)dst, 16, 1024, NTV);
            __builtin_amdgcn_global_load_lds((ps_glb_t *)(base + voff), (ps_lds_t *)dst, 16, 2048, NTV);
            __builtin_amdgcn_global_load_lds((ps_glb_t *)(base + voff), (ps_lds_t *)dst, 16, 3072, NTV);
        } else {
            // the op's last group: pieces past the end re-read its last piece (into ring space nobody reads)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t o = min(voff + (uint32_t)q * 1024u, last);
                __builtin_amdgcn_global_load_lds((ps_glb_t *)(base + o), (ps_lds_t *)(dst + q * 1024), 16, 0, NTV);
            }
        }
        st.G += STEP;
        st.roff = st.roff + STEP >= PS_RING ? st.roff + STEP - PS_RING : st.roff + STEP;
        st.issued++;
        // bounded depth; groups older than the depth have landed (loads return in order).  Published: the stream position
        // of this wave's oldest group that may still be in flight -- every group of this wave below it is in LDS.
        const unsigned thin = ps_asm_lds_ld(st.ctl_addr + 4 * PS_C_GATHER);
        uint32_t behind;
        if (thin != 0u) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(IFA_PS_THIN - 4) : "memory");
            behind = (uint32_t)(IFA_PS_THIN / 4 - 1);
        } else if constexpr (DEEP) {
            asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
            behind = 14u;
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(IFA_PS_INFLIGHT - 4) : "memory");
            behind = (uint32_t)(IFA_PS_INFLIGHT / 4 - 1);
        }
        if (st.issued > behind) {
            const uint32_t landed_to = st.G - behind * STEP;
            if ((int)(landed_to - st.pub) > 0) { st.pub = landed_to; ps_asm_lds_st(st.ctl_addr + 4 * (PS_C_FILLED + st.li), st.pub); }
        }
    }
    st.op_pos = op_end;
}
__device__ __forceinline__ void ps_load_op(PsCtx &c, const uint8_t *base, uint32_t S, PsLoad &st, unsigned code)
{
    if (S == 0u) return;
    // tuning variants (option persist_depth): bit 0 = 60 KiB in flight per loader wave instead of 32, bit 1 = default
    // cache policy instead of nt
    switch (st.variant) {
    case 1: ps_load_op_t<2, true>(c, base, S, st, code); break;
    case 2: ps_load_op_t<0, false>(c, base, S, st, code); break;
    case 3: ps_load_op_t<0, true>(c, base, S, st, code); break;
    default: ps_load_op_t<IFA_PS_NT ? 2 : 0, false>(c, base, S, st, code); break;
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
    // wave roles.  Waves are dealt to the four SIMDs round-robin: with 8 waves, hardware waves 0 and 4 share SIMD 0 -- they
    // are the two loaders -- and the six consumers sit two per SIMD on SIMDs 1-3 (consumers 0-3 = hardware waves 1, 2, 3,
    // 5: still one per ... the attention's 256 threads).  c.w = consumer index, or PS_NC + loader index.
    {
        const int hw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (PS_NC == 6 && PS_NL == 2) c.w = (hw & 3) == 0 ? PS_NC + (hw >> 2) : (hw < 4 ? hw - 1 : hw - 2);
        else c.w = hw;
    }
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

    if (c.w >= PS_NC) {
        // ---------------- loader waves
        PsLoad st;
        st.li = (unsigned)(c.w - PS_NC);
        st.G = st.li * 4096u; st.roff = st.G; st.pub = 0; st.free_to = PS_RING; st.op_pos = 0; st.issued = 0; st.blocked = 0;
        st.free_run = P.tune_depth >= 8; st.variant = P.tune_depth & 3;
        st.ctl_addr = (unsigned)(uintptr_t)(ps_lds_char *)(smem + PS_CTL_OFF);
        if (P.tune_prio) __builtin_amdgcn_s_setprio(3);
        for (int L = P.layer_begin; L < P.layer_end; L++) {
            c.cur_layer = L;
            if (ps_aborted(c)) break;
            const ps_layer_c &ly = ((ps_layer_c *)P.layers)[L];
            long long *ltr = (P.trace && L == P.trace_layer && c.lane == 0 && st.li == 0) ? P.trace + (size_t)c.cu * 32 : nullptr;
            if (ltr) { ltr[26] = wall_clock64(); st.blocked = 0; }
            ps_load_op(c, ly.wqkv + (size_t)g.first[0] * P.row_bytes_a, (uint32_t)g.n[0] * P.row_bytes_a, st, 0x40);
            ps_load_op(c, ly.wo + (size_t)g.first[1] * P.row_bytes_a, (uint32_t)g.n[1] * P.row_bytes_a, st, 0x41);
            if (ltr) ltr[27] = wall_clock64();
            ps_load_op(c, ly.w13 + (size_t)g.first[2] * 2 * P.row_bytes_a, (uint32_t)g.n[2] * 2 * P.row_bytes_a, st, 0x42);
            if (ltr) ltr[28] = wall_clock64();
            ps_load_op(c, ly.w2 + (size_t)g.first[3] * P.row_bytes_b, (uint32_t)g.n[3] * P.row_bytes_b, st, 0x43);
            if (ltr) { ltr[29] = wall_clock64(); ltr[25] = (long long)st.blocked; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ps_asm_lds_st(st.ctl_addr + 4 * (PS_C_FILLED + st.li), st.G);
        return;
    }

    // ---------------- consumers
    if (P.tune_depth >= 8) return;       // tuning: the loader alone (raw rate of the weight stream)
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
    // unit rows per batch: bounded by registers (a row of NJ blocks costs NJ * DW VGPRs) and by code size
    constexpr int UM_A = 4;
    constexpr int UM_B = NJB * FmtB::DW <= 20 ? 4 : 2;
    constexpr int MAXL = 16;             // granule requests in flight per lane of a gathering wave
    // head this CU serves (or -1)
    int my_head = -1;
    {
        const int stride = c.ncu / P.heads;
        const int hc = c.cu / stride;
        if (hc < P.heads && ps_head_cu(hc, P.heads, c.ncu) == c.cu) my_head = hc;
    }
    uint32_t lpos = 0;      // stream position of the current layer's first byte
    // Values the op loop selects between, as opaque scalars.  hipcc folds "cond ? P.a : P.b" over neighbouring fields of
    // the by-value argument block into ONE load with a computed address, which forces a copy of the block into scratch
    // memory -- and then every argument is read back with vector memory loads, each of them waiting behind the
    // write-through granule stores in flight (~2 us): the first compact version of this kernel ran at 100 us per layer.
#define PS_OPAQUE(x) asm volatile("" : "+s"(x))
    unsigned long long *pg_x = P.g_x, *pg_qkv = P.g_qkv, *pg_att = P.g_att, *pg_a = P.g_a, *pg_act = P.g_act;
    PS_OPAQUE(pg_x); PS_OPAQUE(pg_qkv); PS_OPAQUE(pg_att); PS_OPAQUE(pg_a); PS_OPAQUE(pg_act);
    int p_dim = P.dim, p_ffn = P.ffn;
    PS_OPAQUE(p_dim); PS_OPAQUE(p_ffn);
    float p_base_a = P.attn_norm_base, p_base_f = P.ffn_norm_base;
    PS_OPAQUE(p_base_a); PS_OPAQUE(p_base_f);
#undef PS_OPAQUE

#pragma nounroll
    for (int L = P.layer_begin; L < P.layer_end; L++, lpos += g.layer_bytes) {
        c.cur_layer = L;
        if (ps_aborted(c)) break;
        const ps_layer_c &ly = ((ps_layer_c *)P.layers)[L];
        c.trace = (P.trace && L == P.trace_layer && w == 0) ? P.trace + (size_t)c.cu * 32 : nullptr;
        ps_stamp(c, 0);
        if (c.trace && lane == 0) c.trace[30] = clock64();      // shader clock (s_memtime) next to the 100 MHz stamps: the clock the part holds
        // The four weight ops of the layer as ONE loop body with per-op scalars, so that the gather, the quantiser and
        // the row loop exist once in the code:
        //   op 0: x      -> RMSNorm -> Q8 -> wq | wk | wv rows                       -> q | k | v
        //   op 1: [attention on this CU's head]  quantised attention output -> wo rows (+ bias, + residual x) -> a
        //   op 2: a      -> RMSNorm -> Q8 -> w1 / w3 rows -> act(t1) * t2            -> gated product
        //   op 3: gated product -> Q8 -> w2 rows (+ bias, + residual a)              -> next layer's x
#pragma nounroll
        for (int op = 0; op < 4; op++) {
            const bool normed = op == 0 || op == 2;
            PsNormW<MAXG_A> NW;
            const half_t *nw = op == 0 ? ly.attn_norm : ly.ffn_norm, *nb = op == 0 ? ly.attn_norm_b : ly.ffn_norm_b;
            if (normed) NW.request(c, nw, nb, P.dim);
            if (op == 1 && my_head >= 0) ps_attention<HD, KVQ8>(c, P, ly, L, my_head, pos, stage_b);
            // ---- the op's input vector from the CUs that produced it
            const unsigned long long *gin = op == 0 ? pg_x : (op == 1 ? pg_att : (op == 2 ? pg_a : pg_act));
            const int n_in = op == 1 ? q_rows / 4 + q_rows / 16 : (op == 3 ? p_ffn / 2 : p_dim / 2);
            const unsigned ep_in = ps_epoch(L, op == 0 ? PS_E_X : (op == 1 ? PS_E_ATT : (op == 2 ? PS_E_A : PS_E_ACT)));
            uint32_t *dst = op == 1 ? reinterpret_cast<uint32_t *>(img_base) : reinterpret_cast<uint32_t *>(stage);
            int f3, n3; ps_part(n_in, PS_NC, w, f3, n3);
            if (op == 0 && L == P.layer_begin) {
                for (int i = f3 + lane; i < f3 + n3; i += 64) dst[i] = reinterpret_cast<const uint32_t *>(P.x_in)[i];
            } else {
                ps_gather<MAXL>(c, gin, f3, n3, ep_in, dst, 0x10u + (unsigned)op);
            }
            ps_cbar(c, 0x20u + (unsigned)op);
            ps_stamp(c, op == 0 ? 1 : (op == 1 ? 6 : (op == 2 ? 8 : 11)));
            // ---- residual values of the rows this CU owns in the op that adds this vector back (wo: x, w2: a)
            if (normed && w == 0) {
                half_t *res = op == 0 ? res_o : res_2;
                const int rf = op == 0 ? g.first[1] : g.first[3], rn = op == 0 ? g.n[1] : g.n[3];
                for (int i = lane; i < rn; i += 64) res[i] = stage[rf + i];
            }
            // ---- Q8 image of the input
            const int cols = op == 3 ? p_ffn : p_dim;
            const PsImg img = ps_img(img_base, cols);
            if (normed) {
                ps_quantize<true, MAXG_A>(c, stage, P.dim, NW, nw != nullptr, nb != nullptr, op == 0 ? p_base_a : p_base_f, P.eps, img, 0x30u + (unsigned)op);
                ps_stamp(c, op == 0 ? 2 : 9);
            } else if (op == 3) {
                const PsNormW<1> none = {};
                ps_quantize<false, MAXG_B>(c, stage, P.ffn, none, false, false, 0.0f, P.eps, img, 0x33u);
                ps_stamp(c, 12);
            }
            // ---- rows
            const unsigned ep_out = op == 3 ? ps_epoch(L + 1, PS_E_X) : ps_epoch(L, op == 0 ? PS_E_QKV : (op == 1 ? PS_E_A : PS_E_ACT));
            unsigned long long *gout = op == 0 ? pg_qkv : (op == 1 ? pg_a : (op == 2 ? pg_act : pg_x));
            // (selected by value: indexing the geometry arrays with `op` would put them in scratch memory, and a scratch
            // load is a vector memory load whose wait also covers the granule stores in flight)
            const uint32_t op_pos = lpos + (op == 0 ? g.off[0] : (op == 1 ? g.off[1] : (op == 2 ? g.off[2] : g.off[3])));
            const uint32_t op_end = lpos + (op == 0 ? g.off[1] : (op == 1 ? g.off[2] : (op == 2 ? g.off[3] : g.layer_bytes)));
            if (op < 3) {
                typename FmtA::X X;
                X.load(img.codes, img.scale, img.xsum, lane, P.nblk_a);
                const int n_units = op == 0 ? g.n[0] : (op == 1 ? g.n[1] : 2 * g.n[2]);
                const int first_u = op == 0 ? g.first[0] : g.first[1];
                const int ub = ps_units_per_batch(n_units, UM_A, op == 2 ? 4 : 2);
                ps_gemv<DT, NJA, UM_A>(c, X, P.nblk_a, P.row_bytes_a, op_pos, op_end, n_units, ub, 0x40u + (unsigned)op,
                    [&](int u0, int nu, float acc) __attribute__((always_inline)) {
                        const int ul = u0 + min(lane, nu - 1);           // this lane's unit (lanes past the batch: its last)
                        half_t y;
                        int gidx;
                        bool store;
                        if (op == 2) {
                            // units 2r, 2r + 1 = w1 row r, w3 row r: the pair's sums meet in the even lane
                            const int row = g.first[2] + (ul >> 1);
                            const half_t t = dec_bias(acc, ly.b13, 2 * g.first[2] + ul);
                            const half_t t2 = __builtin_bit_cast(half_t, (uint16_t)__shfl_down((uint32_t)__builtin_bit_cast(uint16_t, t), 1));
                            const half_t act = f2h(act_fn(h2f(t), P.act_kind));
                            y = f2h(h2f(act) * h2f(t2));
                            const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, y);
                            const uint32_t other = (uint32_t)__shfl_down(mine, 2);
                            store = lane < nu && (lane & 3) == 0;
                            if (store) ps_publish(gout, (g.first[2] + (u0 >> 1) + (lane >> 1)) >> 1, ep_out, mine | (other << 16));
                        } else {
                            if (op == 0) {
                                y = dec_bias(acc, ly.bqkv, g.first[0] + ul);
                            } else {
                                y = dec_bias(acc, ly.bo, g.first[1] + ul);
                                y = f2h(h2f(res_o[ul]) + h2f(y));
                            }
                            const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, y);
                            const uint32_t other = (uint32_t)__shfl_down(mine, 1);
                            if (lane < nu && (lane & 1) == 0) ps_publish(gout, (first_u + u0 + lane) >> 1, ep_out, mine | (other << 16));
                        }
                    });
            } else {
                typename FmtB::X X;
                X.load(img.codes, img.scale, img.xsum, lane, P.nblk_b);
                const bool last = L + 1 == P.layer_end;
                const int ub = ps_units_per_batch(g.n[3], UM_B, 2);
                ps_gemv<DT, NJB, UM_B>(c, X, P.nblk_b, P.row_bytes_b, op_pos, op_end, g.n[3], ub, 0x43u,
                    [&](int u0, int nu, float acc) __attribute__((always_inline)) {
                        const int ul = u0 + min(lane, nu - 1);
                        half_t y = dec_bias(acc, ly.b2, g.first[3] + ul);
                        y = f2h(h2f(res_2[ul]) + h2f(y));
                        const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, y);
                        const uint32_t other = (uint32_t)__shfl_down(mine, 1);
                        if (last) { if (lane < nu) P.x_out[g.first[3] + ul] = y; }
                        else if (lane < nu && (lane & 1) == 0) ps_publish(gout, (g.first[3] + u0 + lane) >> 1, ep_out, mine | (other << 16));
                    });
            }
            ps_stamp(c, op == 0 ? 3 : (op == 1 ? 7 : (op == 2 ? 10 : 13)));
        }
        if (c.trace && lane == 0) c.trace[31] = clock64();
    }
}

} // namespace ifa
