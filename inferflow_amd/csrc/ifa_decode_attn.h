// ifa_decode_attn.h -- the batch-1 / batched decode attention kernels (k_dec_attn and the keys-split-over-workgroups
// variants), split from ifa_decode_kernels.h so that the per-format GEMV translation units do not depend on them.
#pragma once
#include "ifa_decode_kernels.h"

namespace ifa {

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

// Score chains per key.  The reference's Gemm_Alg2 adds a key's HD products as ONE fp32 chain in d order (gemm.h:83-178); this body
// keeps that order by default (IFA_ATTN_NACC = 1): 128 dependent fma per key, the compiler pads the chain with a wait state per
// step.  Round 5 measured the alternative SURVEY 8(c) allows (fp within tolerance): -DIFA_ATTN_NACC=4 sends the pairs
// (2 i, 2 i + 1) -- exactly as q and the key row hold them in memory -- through v_dot2_f32_f16 into four independent chains,
// added as ((c0 + c1) + (c2 + c3)): 64 instructions per key instead of 128 + 128 wait states.  On MI355X the score phase of the
// fused launch's attention tail went 0.76 -> 0.52 us (profiles/r05_attention_dot2_trace.log) -- 0.24 us of a 38 us layer,
// +0.2 % tokens/s, inside the box-to-box noise -- because a single wave's tail is bound by its ~1000 serial instructions and
// five barriers, not by this chain.  It also parts the fused step from the op-level attention kernels (ifa_attn.hip keeps the
// reference order), i.e. it costs the bit-identity "fused decode == op-by-op decode" that tests/test_gpu_engine.py holds for every
// format.  Not worth that: the order-exact chain stays the default, the dot2 form a build option with its numbers on file.
#ifndef IFA_ATTN_NACC
#define IFA_ATTN_NACC 1
#endif
struct ScoreAcc {
    float c[IFA_ATTN_NACC];
    __device__ __forceinline__ ScoreAcc() {
#pragma unroll
        for (int i = 0; i < IFA_ATTN_NACC; i++) c[i] = 0.0f;
    }
    // elements (2 p, 2 p + 1) of the row: p is a constant after unrolling, so c[] stays in registers
    __device__ __forceinline__ void add2(int p, half2_t q, half2_t k) {
        if constexpr (IFA_ATTN_NACC == 1) {
            c[0] = __builtin_fmaf((float)q[0], (float)k[0], c[0]);
            c[0] = __builtin_fmaf((float)q[1], (float)k[1], c[0]);
        } else {
            c[p % IFA_ATTN_NACC] = __builtin_amdgcn_fdot2(q, k, c[p % IFA_ATTN_NACC], false);
        }
    }
    __device__ __forceinline__ void add(int d, float q, float k) { c[d % IFA_ATTN_NACC] = __builtin_fmaf(q, k, c[d % IFA_ATTN_NACC]); }
    __device__ __forceinline__ float total() const {
        if constexpr (IFA_ATTN_NACC == 1) return c[0];
        else if constexpr (IFA_ATTN_NACC == 2) return c[0] + c[1];
        else if constexpr (IFA_ATTN_NACC == 4) return (c[0] + c[1]) + (c[2] + c[3]);
        else { float t = c[0]; for (int i = 1; i < IFA_ATTN_NACC; i++) t = t + c[i]; return t; }
    }
};

// PB: cache rows requested at kernel entry (before the position is known): the engine passes the bucket (64 / 128 / 256) the
// decode call stays inside, rows past it take the loops' direct loads.  KT (F16 cache, head sizes with a power-of-two number
// of 16-byte pieces): the K rows are requested like the V rows -- 16 bytes per lane, a wave instruction covers whole rows --
// and transposed through an XOR-swizzled LDS tile into the one-key-per-lane registers of the score chain.  One key per lane
// straight from memory is 64 different cache lines per wave instruction: the 256-row prefetch was 4096 line requests per
// CU, ~1.7 us of the kernel before its first stamp (r04 trace: the score phase waited for K, the P.V phase for V).
// FUSED (k_dec_qkv_attn, ifa_decode_qkv_attn.h): the body runs as the tail of the QKV launch on the first 256 threads of the
// head's designated workgroup -- the new token's q | k | v values arrive as {epoch, half} granules written by the workgroups
// that computed those rows (FusedIn), the position is already known (rows past it are not requested), head = `h_in`.
struct DecAttnFusedIn {
    const unsigned long long *gran;     // [rows of q | k | v] granules of this layer
    unsigned epoch;                     // tag of this step
    int pos;                            // position of the new token
    unsigned *err;                      // error word: set when a wait gives up
    long long timeout_ticks;            // wall_clock64 ticks (100 MHz) a wait may take
    // the head's quantised output for the Wo rows computed in the SAME launch (k_dec_qkv_attn<.., WO>), or null: granules
    // [cols / 4] code dwords | [cols / 32] scales | [cols / 32] code sums | [heads] done flags, cols = heads * HD
    unsigned long long *att_gran;
};

// indices into the attention-output granules (see DecAttnFusedIn::att_gran)
__host__ __device__ inline int att_gran_count(int heads, int hd) { return heads * hd / 4 + 2 * (heads * hd / 32) + heads; }

// What a thread requests at entry and keeps until the phases that use it: its key's row (or its pieces of the K tile), its V
// pieces, the step's (cos, sin) pair, the new token's values.  A struct so that the fused launch can REQUEST (PHASE 1) long
// before it COMPUTES (PHASE 2): behind its own weight rows, so that the cache rows arrive with the end of the weight stream.
template <int HD, bool Q8, int PB, bool KT>
struct DecAttnRegs {
    static constexpr int DG = HD / 8, NSPLIT = 256 / DG, VPRE = PB / NSPLIT;
    static constexpr int KBYTES = (HD / 32) * 34;
    static constexpr int KALIGN = HD == 128 ? 8 : (HD == 64 ? 4 : 2);
    uint32_t kreg[Q8 ? 1 : HD / 2];
    uint32_t kq32[(Q8 && KALIGN >= 4) ? KBYTES / 4 : 1];
    uint16_t kq16[(Q8 && KALIGN < 4) ? KBYTES / 2 : 1];
    u32x4 kt[KT ? PB / NSPLIT : 1];
    u32x4 vreg[Q8 ? 1 : VPRE];
    uint32_t vqs[Q8 ? VPRE : 1], vqc[Q8 ? VPRE : 1][2];  // Q8: block scale (half bits) and the 8 codes of this thread's 8 dims (dwords: 16-bit members went to scratch)
    float rope_cs, rope_sn;
    half_t q_in, k_in, v_in;
};

// PHASE 0: the whole kernel; 1: the entry requests only; 2: everything behind them (R filled by a PHASE 1 call)
template <int HD, bool Q8, bool BATCH, int PB, bool KT, bool FUSED, int PHASE = 0>
__device__ __forceinline__ void dec_attn_body(char *smem, const half_t *pq, const uint8_t *pkc, const uint8_t *pvc, int pheads, int pkvh,
                                              const DecAttnParams &P, const int h_in, const DecAttnFusedIn &F, DecAttnRegs<HD, Q8, PB, KT> &R)
{
    int pos_b = 0;
    if constexpr (FUSED) pos_b = F.pos;
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
    static_assert(!KT || (!Q8 && !BATCH && (DG & (DG - 1)) == 0 && PB % NSPLIT == 0), "K tile through LDS: F16 rows, one query, power-of-two pieces per row");
    static_assert(!(FUSED && BATCH), "the fused tail is a single-query kernel");
    static_assert(PB >= NSPLIT && PB <= DEC_ATTN_MIN_ROWS, "prefetch bucket");
    uint8_t *ktile = reinterpret_cast<uint8_t *>(opart + NSPLIT * HD);                         // KT: [PB][HD] halves, 16-byte pieces XOR-swizzled by key
    half_t *S = reinterpret_cast<half_t *>(ktile + (KT ? (size_t)PB * HD * 2 : 0));            // [n_ctx]
    const int h = h_in, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    auto &kreg = R.kreg; auto &kq32 = R.kq32; auto &kq16 = R.kq16; auto &kt = R.kt; auto &vreg = R.vreg; auto &vqs = R.vqs; auto &vqc = R.vqc;
    float &rope_cs = R.rope_cs, &rope_sn = R.rope_sn;
    half_t &q_in = R.q_in, &k_in = R.k_in, &v_in = R.v_in;
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
    if constexpr (PHASE != 2) {
    q_in = (half_t)0; k_in = (half_t)0; v_in = (half_t)0;
    if constexpr (!FUSED) { q_in = pq[(size_t)h * HD + dq]; k_in = pq[(size_t)(pheads + kvh) * HD + dq]; v_in = pq[(size_t)(pheads + pkvh + kvh) * HD + dq]; }
    }
    // (tid < DEC_ATTN_MIN_ROWS <= rows of the cache.)  Batched step: the position arrived with the cache pointers, so rows past
    // the context are clamped to the last one (duplicate addresses: one cache line) -- unclamped, every (head, query) workgroup
    // pulled 2 x 64 KB of cache rows whatever its context: 134 MB per layer at 32 queries, the whole cost of that launch
    const int dg = tid % DG, sp = tid / DG;
    const bool vact = (256 % DG == 0) || sp < NSPLIT;   // this thread takes part in P.V
    constexpr int VPRE = PB / NSPLIT;                   // prefetched V keys per thread: j = sp + NSPLIT*i (PB keys)
    if constexpr (PHASE != 2) {
    if constexpr (KT) {
#pragma unroll
        for (int i = 0; i < PB / NSPLIT; i++) kt[i] = IFA_GP(u32x4, pkc + (size_t)min(sp + NSPLIT * i, FUSED ? min(pos_b, PB - 1) : PB - 1) * row_bytes + head_off)[dg];
    } else {
        load_k((BATCH || FUSED) ? min(tid, min(pos_b, PB - 1)) : min(tid, PB - 1));
    }
    // this step's (cos, sin) pair of the thread that will rotate: requested BETWEEN the K and the V rows -- behind both it was the
    // newest request, and the rotation waited (vmcnt(0)) for the whole prefetch; unconditional (a valid dummy address without RoPE)
    rope_cs = 1.0f; rope_sn = 0.0f;
    {
        const int c = min(tid < HD / 2 ? tid : tid - HD / 2, HD / 2 - 1);
        const float *rt = rope_tab ? rope_tab : reinterpret_cast<const float *>(pq);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 cs = *IFA_GP(f32x2, rt + 2 * c);
        if (P.rope_order != 0) { rope_cs = cs[0]; rope_sn = cs[1]; }
    }
    const size_t vq_off = head_off + (size_t)(dg / 4) * 34;
#pragma unroll
    for (int i = 0; i < VPRE; i++) {
        const int j = min(sp + NSPLIT * i, BATCH ? pos_b : (FUSED ? min(pos_b, PB - 1) : PB - 1));
        if constexpr (!Q8) {
            vreg[i] = IFA_GP(u32x4, pvc + (size_t)j * row_bytes + head_off)[dg];
        } else {
            // the thread's 8 codes as ONE 8-byte request at a 2-byte-aligned address (global memory takes unaligned dwords): five
            // 2-byte requests per key were 80 load instructions per thread ahead of everything else in the kernel
            const auto *blk = IFA_GP(uint16_t, pvc + (size_t)j * row_bytes + vq_off);
            typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
            vqs[i] = blk[0];
            const u32x2_a2 cw = *IFA_GP(u32x2_a2, blk + 1 + (dg % 4) * 4);
            vqc[i][0] = cw[0]; vqc[i][1] = cw[1];
        }
    }
    }       // (PHASE != 2)
    if constexpr (PHASE == 1) return;

    // ---- everything below may wait for the argument block: the position, this step's (cos, sin) pair of the thread that
    // will rotate (requested now, used after the staging barrier)
    const bool tr = P.trace != nullptr && tid == 0;
    if (tr) P.trace[h * 8 + 0] = wall_clock64();
    // (a SCALAR load through the constant address space: as a vector load it was the newest request of the wave, and waiting for it
    //  -- vmcnt(0) -- meant waiting for every K / V row requested above before the new token's values could even be staged)
    const int pos = (BATCH || FUSED) ? pos_b : *(const __attribute__((address_space(4))) int *)(P.state + 1);
    const int n_ctx = pos + 1;
    if constexpr (FUSED) {
        // the new token's values: one granule per row, valid when its tag is this step's epoch.  The producers are the other
        // workgroups of the head's group (and this one): bounded wait, the error word tells the host
        if (tid < HD) {
            const unsigned long long *gq = F.gran + (size_t)h * HD + tid, *gk = F.gran + (size_t)(pheads + kvh) * HD + tid,
                                     *gv = F.gran + (size_t)(pheads + pkvh + kvh) * HD + tid;
            const long long t_give_up = wall_clock64() + F.timeout_ticks;
            for (;;) {
                const unsigned long long a = __hip_atomic_load(gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long b = __hip_atomic_load(gk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long c = __hip_atomic_load(gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (unsigned)(a >> 32) == F.epoch && (unsigned)(b >> 32) == F.epoch && (unsigned)(c >> 32) == F.epoch;
                q_in = __builtin_bit_cast(half_t, (uint16_t)a); k_in = __builtin_bit_cast(half_t, (uint16_t)b); v_in = __builtin_bit_cast(half_t, (uint16_t)c);
                if (__all(ok)) break;
                if (wall_clock64() > t_give_up) { if (lane == 0) atomicExch(F.err, 0x51u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    // ---- stage q, k_new, v_new; RoPE on q and k (TensorOpr::PositionEmbedding, F16 in/out)
    if (tid < HD) { qs[tid] = q_in; kn[tid] = k_in; vn[tid] = v_in; }
    if constexpr (KT) {
        // the K rows into the tile (piece dg of key j at piece slot dg ^ (j % DG): a 16-lane group writes one whole row, and
        // reads -- key j = lane, piece i -- hit 16 different slots) under the SAME barrier as the staged values
#pragma unroll
        for (int i = 0; i < PB / NSPLIT; i++) {
            const int j = sp + NSPLIT * i;
            *reinterpret_cast<u32x4 *>(ktile + ((size_t)j * DG + (size_t)(dg ^ (j & (DG - 1)))) * 16) = kt[i];
        }
    }
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

    if constexpr (KT) {
        // key tid's row out of the tile (written before the staging barrier) into the score chain's registers
        if (tid < PB) {
#pragma unroll
            for (int i = 0; i < DG; i++) {
                const u32x4 t = *reinterpret_cast<const u32x4 *>(ktile + ((size_t)tid * DG + (size_t)(i ^ (tid & (DG - 1)))) * 16);
                kreg[4 * i] = t[0]; kreg[4 * i + 1] = t[1]; kreg[4 * i + 2] = t[2]; kreg[4 * i + 3] = t[3];
            }
        }
    }
    // ---- scores: one key per lane, fp32 fma in d order (Gemm_Alg2_Kernel order, products exact)
    if (tr) P.trace[h * 8 + 2] = wall_clock64();
    const float alpha = 1.0f / sqrtf((float)HD) / P.kq_scale;
    const float mk = P.alibi ? alibi_slope(h + P.alibi_base, P.alibi_total) : 0.0f;
    float lmax = -INFINITY;
    // BATCH (round 6, second session): the key row of the NEXT iteration is requested (into a second register set, clamped, unconditional)
    // before this iteration's products -- the loop asked for a row and waited for it, one round trip per 256 keys with one workgroup per
    // (head, query) and CU.  Single-query kernels stay as they were (they hand contexts past 320 keys to the split kernels).
    uint32_t kreg2[(BATCH && !Q8) ? HD / 2 : 1];
    uint32_t kq32b[(BATCH && Q8 && KALIGN >= 4) ? KBYTES / 4 : 1];
    constexpr bool KPRE = BATCH && (!Q8 || KALIGN >= 4);
    bool have = false;                                   // (KPRE) kreg / kq32 already hold row j
    for (int j = tid; j < n_ctx; j += 256) {
        ScoreAcc acc;
        if constexpr (KPRE) {
            const uint8_t *rowp2 = pkc + (size_t)min(j + 256, n_ctx - 1) * row_bytes + head_off;
            if constexpr (!Q8) {
#pragma unroll
                for (int i = 0; i < HD / 8; i++) {
                    const u32x4 t = IFA_GP(u32x4, rowp2)[i];
                    kreg2[4 * i] = t[0]; kreg2[4 * i + 1] = t[1]; kreg2[4 * i + 2] = t[2]; kreg2[4 * i + 3] = t[3];
                }
            } else if constexpr (KALIGN == 8) {
#pragma unroll
                for (int i = 0; i < KBYTES / 8; i++) { const u32x2 t = IFA_GP(u32x2, rowp2)[i]; kq32b[2 * i] = t[0]; kq32b[2 * i + 1] = t[1]; }
            } else {
#pragma unroll
                for (int i = 0; i < KBYTES / 4; i++) kq32b[i] = IFA_GP(uint32_t, rowp2)[i];
            }
        }
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
            for (int i = 0; i < HD / 2; i++) acc.add2(i, reinterpret_cast<const half2_t *>(qs)[i], __builtin_bit_cast(half2_t, knr[i]));
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
                } else if (j >= PB && !have) load_k(j);       // rows past the prefetched bucket: load now
            } else if (j >= PB && !have) load_k(j);
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
                            acc.add2(2 * w4, reinterpret_cast<const half2_t *>(qs)[b * 16 + 2 * w4], lo);
                            acc.add2(2 * w4 + 1, reinterpret_cast<const half2_t *>(qs)[b * 16 + 2 * w4 + 1], hi);
                        }
                    } else {
                        const float sc = hbits2f(scb);
#pragma unroll
                        for (int i = 0; i < 32; i++) {
                            const int qv = (int)(int8_t)kbyte(b * 34 + 2 + i);
                            const float kvv = h2f(f2h((float)qv * sc));
                            acc.add(i, h2f(qs[b * 32 + i]), kvv);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < HD / 2; i++) acc.add2(i, reinterpret_cast<const half2_t *>(qs)[i], __builtin_bit_cast(half2_t, kreg[i]));
            }
        }
        const float c = acc.total();
        half_t s = f2h(alpha * c);
        if (P.alibi) { float a = (float)j * mk; s = f2h(a + h2f(s)); }
        S[j] = s;
        lmax = fmaxf(lmax, P.kq_scale * h2f(s));
        if constexpr (KPRE) {                            // the next iteration's row moves into the working registers
            if constexpr (!Q8) {
#pragma unroll
                for (int i = 0; i < HD / 2; i++) kreg[i] = kreg2[i];
            } else {
#pragma unroll
                for (int i = 0; i < KBYTES / 4; i++) kq32[i] = kq32b[i];
            }
            have = true;
        }
    }
    if (tr) P.trace[h * 8 + 3] = wall_clock64();
    lmax = wave_max(lmax);
    if (n_ctx <= 64) {
        // all keys live in wave 0: its maximum IS the row maximum (the other waves hold -inf) and its sum the row sum (the others
        // add +0, exact), so the two cross-wave reductions and their barriers fall away -- the same values bit for bit
        if (wave == 0) {
            if (tr) P.trace[h * 8 + 4] = wall_clock64();
            float e = 0.0f;
            if (tid < n_ctx) e = expf(P.kq_scale * h2f(S[tid]) - lmax);
            const float lsum0 = wave_sum(e);
            const float inv0 = 1.0f / (((lsum0 + 0.0f) + 0.0f) + 0.0f);
            if (tid < n_ctx) S[tid] = f2h(h2f(f2h(e)) * inv0);
        }
        __syncthreads();
    } else {
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
    }
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
            else if constexpr (Q8) acc_q8w(pj, (uint16_t)vqs[i], (uint16_t)(vqc[i][0] & 0xFFFFu), (uint16_t)(vqc[i][0] >> 16), (uint16_t)(vqc[i][1] & 0xFFFFu), (uint16_t)(vqc[i][1] >> 16));
            else acc_v(pj, vreg[i]);
        }
    }
    // keys past the prefetched bucket (round 6, second session): batches of eight keys of this thread's sequence, the next batch
    // requested before the current one is multiplied (unconditional, clamped rows).  The loop asked for one V piece per iteration and
    // waited for it: a batched step at 2048 keys spent 96 us per layer in this kernel.  Same order of accumulation as the plain loop.
    {
        constexpr int TB = 8;
        u32x4 tcur[Q8 ? 1 : TB], tnxt[Q8 ? 1 : TB];
        uint32_t ts_c[Q8 ? TB : 1], tc_c[Q8 ? TB : 1][2], ts_n[Q8 ? TB : 1], tc_n[Q8 ? TB : 1][2];
        const size_t tq_off = head_off + (size_t)(dg / 4) * 34;
        auto tload = [&](u32x4 (&dst)[Q8 ? 1 : TB], uint32_t (&ds)[Q8 ? TB : 1], uint32_t (&dc)[Q8 ? TB : 1][2], int j) {
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int jr = min(j + u * NSPLIT, n_ctx - 1);
                if constexpr (!Q8) dst[u] = IFA_GP(u32x4, pvc + (size_t)jr * row_bytes + head_off)[dg];
                else {
                    const auto *blk = IFA_GP(uint16_t, pvc + (size_t)jr * row_bytes + tq_off);
                    typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
                    ds[u] = blk[0];
                    const u32x2_a2 cw = *IFA_GP(u32x2_a2, blk + 1 + (dg % 4) * 4);
                    dc[u][0] = cw[0]; dc[u][1] = cw[1];
                }
            }
        };
        int j = vact ? sp + NSPLIT * VPRE : n_ctx;
        if (j < n_ctx) tload(tcur, ts_c, tc_c, j);
        for (; j < n_ctx; j += TB * NSPLIT) {
            tload(tnxt, ts_n, tc_n, j + TB * NSPLIT);
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int jj = j + u * NSPLIT;
                if (jj >= n_ctx) break;
                const float pj = h2f(S[jj]);
                if (jj == pos) acc_new(pj);
                else if constexpr (Q8) acc_q8w(pj, (uint16_t)ts_c[u], (uint16_t)(tc_c[u][0] & 0xFFFFu), (uint16_t)(tc_c[u][0] >> 16), (uint16_t)(tc_c[u][1] & 0xFFFFu), (uint16_t)(tc_c[u][1] >> 16));
                else acc_v(pj, tcur[u]);
            }
#pragma unroll
            for (int u = 0; u < TB; u++) {
                if constexpr (Q8) { ts_c[u] = ts_n[u]; tc_c[u][0] = tc_n[u][0]; tc_c[u][1] = tc_n[u][1]; }
                else tcur[u] = tnxt[u];
            }
        }
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
        if constexpr (FUSED && HD % 32 == 0) {
            if (F.att_gran) {
                // the Alg2 quantiser of dec_attn_emit_q8 (same expressions), published as granules: four codes per dword (the quad's
                // codes gathered with two DPP quad permutes), the block's scale and code sum by its first lane, the head's flag last
                const float val = h2f(yh);
                const float mxv = half_wave_max(fabsf(val));
                const float qsc = mxv / 127;
                const int qv = q8_round_div1(val, qsc);
                const int sum = half_wave_sum_i32(qv);
                const int q1 = dpp_xor1(qv), q2 = dpp_xor2(qv), q3 = dpp_xor1(q2);
                const unsigned long long tagw = (unsigned long long)F.epoch << 32;
                const int cols = pheads * HD, nc = cols / 4, nbk = cols / 32;
                if ((tid & 3) == 0) {
                    const uint32_t w4 = (uint32_t)(qv & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
                    __hip_atomic_store(F.att_gran + (h * HD + tid) / 4, tagw | w4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if ((tid & 31) == 0) {
                    const int blk = (h * HD + tid) >> 5;
                    __hip_atomic_store(F.att_gran + nc + blk, tagw | __builtin_bit_cast(uint32_t, h2f(f2h(qsc))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(F.att_gran + nc + nbk + blk, tagw | __builtin_bit_cast(uint32_t, (float)sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tid == 0) __hip_atomic_store(F.att_gran + nc + 2 * nbk + h, tagw | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (tr) P.trace[h * 8 + 7] = wall_clock64();
}

template <int HD, bool Q8, bool BATCH = false, int PB = DEC_ATTN_MIN_ROWS, bool KT = false>
__global__ void __launch_bounds__(256) k_dec_attn(const half_t *pq, const uint8_t *pkc, const uint8_t *pvc, int pheads, int pkvh,
                                                  const DecAttnParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DecAttnRegs<HD, Q8, PB, KT> R;
    dec_attn_body<HD, Q8, BATCH, PB, KT, false>(smem, pq, pkc, pvc, pheads, pkvh, P, (int)blockIdx.x, DecAttnFusedIn{}, R);
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
    // Round 6 (second session): the key rows of the first pass are requested BEFORE the new token's values are staged, rotated and
    // stored (they depend on the split only; the new token's own row is never read from the cache), and the rows of the next pass before
    // the current one is multiplied: the kernel was a chain of ~6 dependent round trips for one or two passes of 256 keys.  Q8 rows come
    // as 4- / 8-byte requests of the head's slice (68 two-byte loads per key before).  Same products in the same order.
    constexpr int KCH = HD / 8;                                   // 16-byte chunks per F16 row
    constexpr int KBYTES = (HD / 32) * 34;                        // Q8: bytes of a head's slice of a row
    constexpr int KALIGN = HD == 128 ? 8 : (HD == 64 ? 4 : 2);    // ... and its alignment (see dec_attn_body)
    constexpr bool QW = Q8 && KALIGN >= 4;                        // wide requests for the Q8 slice
    u32x4 kin[Q8 ? 1 : KCH];
    uint32_t kq[QW ? KBYTES / 4 : 1];
    auto load_pass = [&](u32x4 (&dst)[Q8 ? 1 : KCH], uint32_t (&dq)[QW ? KBYTES / 4 : 1], int jb) {
        if constexpr (!Q8) {
#pragma unroll
            for (int i2 = 0; i2 < KCH; i2++) {                    // piece idx = tid + 256 i2: row idx / KCH, chunk idx % KCH
                const int idx = tid + 256 * i2;
                const int jr = min(jb + idx / KCH, max(j1 - 1, 0));
                dst[i2] = reinterpret_cast<const u32x4 *>(P.kcache + (size_t)jr * row_bytes + head_off)[idx % KCH];
            }
        } else if constexpr (QW) {
            const uint8_t *rowp = P.kcache + (size_t)min(jb + tid, max(j1 - 1, 0)) * row_bytes + head_off;
            if constexpr (KALIGN == 8) {
#pragma unroll
                for (int i = 0; i < KBYTES / 8; i++) { const u32x2 t = reinterpret_cast<const u32x2 *>(rowp)[i]; dq[2 * i] = t[0]; dq[2 * i + 1] = t[1]; }
            } else {
#pragma unroll
                for (int i = 0; i < KBYTES / 4; i++) dq[i] = reinterpret_cast<const uint32_t *>(rowp)[i];
            }
        }
    };
    // the new token's values and this step's (cos, sin) pair FIRST, the key rows behind them: requests return in issue order, so the
    // staging below waits for a few bytes while the rows are still in flight (all unconditional, clamped: exact vmcnt counting)
    static_assert(HD <= 256, "one value per thread");
    const int dq = min(tid, HD - 1);
    const half_t q_in = P.q[(size_t)h * HD + dq], k_in = P.k_new[(size_t)kvh * HD + dq], v_in = P.v_new[(size_t)kvh * HD + dq];
    float rope_c = 1.0f, rope_s = 0.0f;
    {
        const int c = min(tid < HD / 2 ? tid : tid - HD / 2, HD / 2 - 1);
        const float *rt = P.rope_tab ? P.rope_tab : reinterpret_cast<const float *>(P.q);      // (a valid dummy address without RoPE)
        const float c0v = rt[2 * c], s0v = rt[2 * c + 1];
        if (P.rope_order != 0) { rope_c = c0v; rope_s = s0v; }
    }
    load_pass(kin, kq, j0);
    if (tid < HD) { qs[tid] = q_in; kn[tid] = k_in; vn[tid] = v_in; }
    __syncthreads();
    if (P.rope_order != 0) {
        if (tid < HD) {
            const int c = tid < HD / 2 ? tid : tid - HD / 2;
            rope_apply(tid < HD / 2 ? qs : kn, c, rope_c, rope_s, P.rope_order, P.rope_cols);
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
    // the rotated q in registers (HD halves per thread, read once in 16-byte pieces): the dot products below read q[d] from LDS once per
    // product before -- HD broadcast reads per key and thread, as many LDS instructions as fmas.  Same values, same order.
    half8_t qreg[HD / 8];
#pragma unroll
    for (int i = 0; i < HD / 8; i++) qreg[i] = *reinterpret_cast<const half8_t *>(qs + 8 * i);
    const float alpha = 1.0f / sqrtf((float)HD) / P.kq_scale;
    const float mk = P.alibi ? alibi_slope(h + P.alibi_base, P.alibi_total) : 0.0f;
    float lmax = -INFINITY;
    // F16 cache: a lane's key row (HD halfs) is requested COALESCED -- a wave request covers 64 / (HD/8) whole rows -- and
    // turned through LDS so that every lane then holds its own key's row: one key per lane with its own 16-byte requests
    // touched 64 rows per request (14 us per layer at 4096 keys).  The dot product below is unchanged (fp32 fma in d order).
    constexpr int KROWB = HD * 2 + 16;                            // bytes per staged row (+16: conflict-free row reads)
    extern __shared__ __attribute__((aligned(16))) char kst[];    // [256][KROWB] (F16 cache only: dec_attn_scores_smem)
    for (int jb = j0; jb < j1; jb += 256) {
        const int j = jb + tid;
        // the rows of the NEXT pass are requested as soon as this pass's registers are free (behind the LDS stores / the register copy):
        // with one workgroup per CU (8 splits per head) nothing else overlaps a pass's memory round trip with the products of the one before
        const int jn = min(jb + 256, max(j1 - 1, 0));              // (past the split: clamped, harmless)
        u32x4 krow[Q8 ? 1 : KCH];
        uint32_t kqc[QW ? KBYTES / 4 : 1];
        if constexpr (!Q8) {
            __syncthreads();                                      // the previous pass's rows have been read
#pragma unroll
            for (int i2 = 0; i2 < KCH; i2++) {
                const int idx = tid + 256 * i2;
                *reinterpret_cast<u32x4 *>(kst + (size_t)(idx / KCH) * KROWB + (size_t)(idx % KCH) * 16) = kin[i2];
            }
            load_pass(kin, kq, jn);
            __syncthreads();
#pragma unroll
            for (int i2 = 0; i2 < KCH; i2++) krow[i2] = *reinterpret_cast<const u32x4 *>(kst + (size_t)tid * KROWB + (size_t)i2 * 16);
        } else if constexpr (QW) {
#pragma unroll
            for (int i = 0; i < KBYTES / 4; i++) kqc[i] = kq[i];
            load_pass(kin, kq, jn);
        }
        if (j >= j1) continue;
        float c = 0.0f;
        if (j == pos) {
#pragma unroll 8
            for (int d = 0; d < HD; d++) c = __builtin_fmaf(h2f(qs[d]), h2f(kn[d]), c);
        } else {
            const uint8_t *rowp = P.kcache + (size_t)j * row_bytes + head_off;
            if constexpr (QW) {
                // byte B (compile-time) of the slice held in registers; block b = bytes [34 b, 34 b + 34): scale (half), 32 codes
                auto kb = [&](int B) -> uint32_t { return (kqc[B >> 2] >> (8 * (B & 3))) & 0xFFu; };
#pragma unroll
                for (int b = 0; b < HD / 32; b++) {
                    const float sc = hbits2f((uint16_t)(kb(34 * b) | (kb(34 * b + 1) << 8)));
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint32_t two = kb(34 * b + 2 + 2 * i) | (kb(34 * b + 3 + 2 * i) << 8);
                        const float k0 = h2f(f2h((float)(int)(int8_t)(two & 0xFF) * sc)), k1 = h2f(f2h((float)(int)(int8_t)(two >> 8) * sc));
                        c = __builtin_fmaf((float)qreg[(b * 32 + 2 * i) >> 3][(2 * i) & 7], k0, c);
                        c = __builtin_fmaf((float)qreg[(b * 32 + 2 * i + 1) >> 3][(2 * i + 1) & 7], k1, c);
                    }
                }
            } else if constexpr (Q8) {
#pragma unroll
                for (int b = 0; b < HD / 32; b++) {
                    const uint16_t *p16 = reinterpret_cast<const uint16_t *>(rowp + b * 34);
                    const float sc = hbits2f(p16[0]);
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint32_t two = p16[1 + i];
                        const float k0 = h2f(f2h((float)(int)(int8_t)(two & 0xFF) * sc)), k1 = h2f(f2h((float)(int)(int8_t)(two >> 8) * sc));
                        c = __builtin_fmaf((float)qreg[(b * 32 + 2 * i) >> 3][(2 * i) & 7], k0, c);
                        c = __builtin_fmaf((float)qreg[(b * 32 + 2 * i + 1) >> 3][(2 * i + 1) & 7], k1, c);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < HD / 8; i++) {
                    const half8_t k8 = __builtin_bit_cast(half8_t, krow[i]);
#pragma unroll
                    for (int e = 0; e < 8; e++) c = __builtin_fmaf((float)qreg[i][e], (float)k8[e], c);
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

// Round 6 (second session): the same sums in the same order, fewer memory round trips.  The kernel was a chain of dependent requests --
// the full-row sum walked S with one 2-byte load per thread and iteration (17 trips at 4096 keys), the P.V loop kept four V rows per
// thread in flight (five trips per 320-key split), the Q8 loop one key -- 21 us per layer at 4096 keys for 33 MB of V (1.6 TB/s).  Now the
// score row is staged through LDS in 16-byte pieces, 4096 keys at a time (thread t still adds keys t, t + 256, ... in ascending order),
// and the V rows of the next eight keys of a thread are requested before the current eight are multiplied.
constexpr int DEC_PV_STAGE = 4096;      // keys of the score row staged per pass (multiple of 256)

template <int HD, bool Q8>
__global__ void __launch_bounds__(256) k_dec_attn_pv(const DecAttnParams P, const DecAttnSplitWs ws)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DG = HD / 8, NSPLIT = 256 / DG;
    float *red = reinterpret_cast<float *>(smem);                    // [8]
    float *opart = red + 8;                                          // [NSPLIT][HD]
    half_t *Sst = reinterpret_cast<half_t *>(opart + NSPLIT * HD);   // [DEC_PV_STAGE + 8] staged piece of the score row
    half_t *Pl = Sst + DEC_PV_STAGE + 8;                             // this split's probabilities
    const int h = blockIdx.x, sidx = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = *(const __attribute__((address_space(4))) int *)(P.state + 1), n_ctx = pos + 1;      // (scalar load: see k_dec_attn)
    int j0, j1; dec_split_range(n_ctx, ws.nsplits, sidx, j0, j1);
    const int group = P.heads / P.kv_heads, kvh = h / group;
    const int kv_dim = P.kv_heads * HD;
    const size_t row_bytes = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
    const size_t head_off = Q8 ? (size_t)((kvh * HD) / 32) * 34 : (size_t)kvh * HD * 2;
    const half_t *Sg = ws.S + (size_t)h * P.max_ctx;
    const int dg = tid % DG, sp = tid / DG;
    const bool vact = (256 % DG == 0) || sp < NSPLIT;
    // ---- the first V rows of this thread's key sequence go out before anything else (they depend on the split only)
    constexpr int VB = 8;                                             // keys per batch and thread
    const int step = NSPLIT;
    u32x4 vcur[Q8 ? 1 : VB], vnxt[Q8 ? 1 : VB];
    uint32_t qs_c[Q8 ? VB : 1], qc_c[Q8 ? VB : 1][2], qs_n[Q8 ? VB : 1], qc_n[Q8 ? VB : 1][2];      // Q8: block scale (half bits) + this thread's 8 codes
    const size_t vq_off = head_off + (size_t)(dg / 4) * 34;
    auto load_v = [&](u32x4 (&dst)[Q8 ? 1 : VB], uint32_t (&ds)[Q8 ? VB : 1], uint32_t (&dc)[Q8 ? VB : 1][2], int j) {
#pragma unroll
        for (int u = 0; u < VB; u++) {
            const int jr = min(j + u * step, max(j1 - 1, 0));
            if constexpr (!Q8) dst[u] = reinterpret_cast<const u32x4 *>(P.vcache + (size_t)jr * row_bytes + head_off)[dg];
            else {
                // the thread's 8 codes as ONE 8-byte request at a 2-byte-aligned address (see dec_attn_body)
                const uint16_t *blk = reinterpret_cast<const uint16_t *>(P.vcache + (size_t)jr * row_bytes + vq_off);
                typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
                ds[u] = blk[0];
                const u32x2_a2 cw = *reinterpret_cast<const u32x2_a2 *>(blk + 1 + (dg % 4) * 4);
                dc[u][0] = cw[0]; dc[u][1] = cw[1];
            }
        }
    };
    const int jv0 = vact ? j0 + sp : j1;
    if (jv0 < j1) load_v(vcur, qs_c, qc_c, jv0);
    float mx = -INFINITY;
    for (int s2 = 0; s2 < ws.nsplits; s2++) mx = fmaxf(mx, ws.lmax[h * ws.nsplits + s2]);
    // the full-row sum, in the same order as the one-workgroup kernel (strided by 256, wave tree, 4 waves); the row comes through LDS
    float lsum = 0.0f;
    for (int c0 = 0; c0 < n_ctx; c0 += DEC_PV_STAGE) {
        const int cn = min(DEC_PV_STAGE, n_ctx - c0);
        const size_t g0 = (size_t)h * P.max_ctx + c0;                // first half of this piece in the workspace
        const int off = (int)(g0 & 7);                                // ... and its distance from a 16-byte boundary
        const u32x4 *src = reinterpret_cast<const u32x4 *>(ws.S + (g0 - off));
        u32x4 piece[DEC_PV_STAGE / 8 / 256 + 1];
#pragma unroll
        for (int i2 = 0; i2 < DEC_PV_STAGE / 8 / 256 + 1; i2++) {
            const int pc = tid + 256 * i2;
            if (pc * 8 < cn + off) piece[i2] = src[pc];
        }
        __syncthreads();                                              // (the previous piece has been read)
#pragma unroll
        for (int i2 = 0; i2 < DEC_PV_STAGE / 8 / 256 + 1; i2++) {
            const int pc = tid + 256 * i2;
            if (pc * 8 < cn + off) *reinterpret_cast<u32x4 *>(Sst + (size_t)pc * 8) = piece[i2];
        }
        __syncthreads();
        for (int j = c0 + tid; j < c0 + cn; j += 256) lsum += expf(P.kq_scale * h2f(Sst[j - c0 + off]) - mx);
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / (((red[0] + red[1]) + red[2]) + red[3]);
    for (int j = j0 + tid; j < j1; j += 256) {
        const half_t eh = f2h(expf(P.kq_scale * h2f(Sg[j]) - mx));
        Pl[j - j0] = f2h(h2f(eh) * inv);
    }
    __syncthreads();
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = 0.0f;
    // batches of VB keys of this thread's sequence (j, j + NSPLIT, ...): the next batch is requested before the current one is
    // multiplied; rows past the split are clamped and skipped -- the order of accumulation is that of a plain loop
    for (int j = jv0; j < j1; j += VB * step) {
        const bool more = j + VB * step < j1;
        if (more) load_v(vnxt, qs_n, qc_n, j + VB * step);
#pragma unroll
        for (int u = 0; u < VB; u++) {
            if (j + u * step >= j1) break;
            const float pu = h2f(Pl[j + u * step - j0]);
            if constexpr (Q8) {
                const float sc = hbits2f((uint16_t)qs_c[u]);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t two = (qc_c[u][e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
                    o[2 * e] = __builtin_fmaf(pu, h2f(f2h((float)(int)(int8_t)(two & 0xFF) * sc)), o[2 * e]);
                    o[2 * e + 1] = __builtin_fmaf(pu, h2f(f2h((float)(int)(int8_t)(two >> 8) * sc)), o[2 * e + 1]);
                }
            } else {
                const half8_t v8 = __builtin_bit_cast(half8_t, vcur[u]);
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = __builtin_fmaf(pu, (float)v8[e], o[e]);
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < VB; u++) {
                if constexpr (Q8) { qs_c[u] = qs_n[u]; qc_c[u][0] = qc_n[u][0]; qc_c[u][1] = qc_n[u][1]; }
                else vcur[u] = vnxt[u];
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
    return 8 * 4 + nsplit * head_dim * 4 + (size_t)(DEC_PV_STAGE + 8) * 2 + chunk * 2 + 16;      // + the staged piece of the score row
}

__host__ __device__ inline size_t dec_attn_smem(int head_dim, int max_ctx, int ktile_rows = 0)
{
    const size_t nsplit = 256 / (head_dim / 8);
    return (size_t)head_dim * 3 * 2 + 16 * 4 + nsplit * head_dim * 4 + (size_t)ktile_rows * head_dim * 2 + (((size_t)max_ctx * 2 + 15) & ~(size_t)15) + 16;
}


} // namespace ifa
