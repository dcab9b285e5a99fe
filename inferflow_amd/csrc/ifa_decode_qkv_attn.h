// ifa_decode_qkv_attn.h -- round 4: the decode attention as the TAIL of the Wq|Wk|Wv launch (one kernel boundary per
// layer less: 5 -> 4 launches).
//
// Reference ops fused here (batch 1): RmsNorm -> Quantize -> Wq, Wk, Wv GEMV (+bias) (inference_worker.cc:1407-1621) and,
// for each query head, RoPE -> SetK/VRows -> GetK/VRows -> Gemm_Alg2 -> SoftMax -> Gemm_Alg2 -> Quantize (:983-1312),
// with the arithmetic and rounding points of k_dec_gemv / k_dec_attn (both bodies are shared: fused == five-launch step bit
// for bit, tests/test_gpu_fused_attn.py).
//
// Why it is legal without a device-wide hand-off: the q / k / v rows of ONE kv group (its `group` query heads + its key head
// + its value head: (group + 2) * HD rows) are a property of the row -> workgroup mapping.  Here the grid is cut into
// kv_heads sets of `gk` workgroups; set g computes exactly the rows of kv group g (k_dec_gemv deals rows round-robin over
// all waves instead), so the attention of a head waits for gk workgroups -- 8 for Llama-2-7B -- not for the grid:
//   * every finished row is published as ONE 8-byte {epoch, half} granule (agent-scope relaxed store: write-through,
//     MI355X_MICROARCH.md "Persistent kernels" price list, form R2 -- no flag, no fence, valid iff the tag matches);
//   * the head's designated workgroup (its own rows are part of the set, so it arrives with the others) requests the
//     head's K / V rows -- now that the position is known, only rows that exist -- polls the 3 * HD granules while those
//     are in flight, and runs dec_attn_body<FUSED>;
//   * tag = (decode-call counter, position of the step): every granule of a layer is rewritten every step, so the tag only has
//     to differ from the previous step's; the arena (one region per layer) is zeroed once at allocation, tags are never 0.
// Every wait is bounded (timeout -> error word -> ifa_model_decode fails loudly); the grid is one workgroup per CU, so all
// producers of a head are resident whenever its consumer waits.
#pragma once
#include "ifa_decode_kernels.h"
#include "ifa_decode_attn.h"

namespace ifa {

struct DecQkvAttnExtra {
    unsigned long long *gran;       // this layer's granules, [(heads + 2 kv_heads) * HD]
    const unsigned *epoch;          // device word: number of the decode call (ifa_model_decode writes it before the steps)
    unsigned epoch_add;             // added to it (ifa_model_time_kernel: distinct tags for repeated launches of one step)
    unsigned *err;                  // error word (ifa_model::ps_err)
    int timeout_us;
    int gk;                         // workgroups per kv group; grid = kv_heads * gk
    int unload;                     // 1: the UL kernels -- the heads' workgroups take no q | k | v rows (256-row bucket, see k_dec_qkv_attn)
};

// IFA_QA_EARLY_KV (measurement, default 0): the head's K / V rows of the entry bucket requested by the attention waves right behind
// their own weight rows (PHASE 1 of dec_attn_body; PHASE 2 at the tail) instead of after the workgroup's last q | k | v row.
// Bit-identical and without effect (150 / 250 keys 734 / 694 against 729 / 689 tok/s, Q8 cache slightly slower): the bytes come through
// the same compute unit either way -- what helps is taking the unit's weight rows away (UL below).  profiles/r06_attn_unload.log
// IFA_QA_WARM_ROWS (default 1): see the UL branch of the kernel
#ifndef IFA_QA_EARLY_KV
#define IFA_QA_EARLY_KV 0
#endif
#ifndef IFA_QA_WARM_ROWS
#define IFA_QA_WARM_ROWS 1
#endif
constexpr int QA_THREADS = 512;     // 8 waves: two per SIMD, 256 registers each (the attention tail needs ~140)

// UL ("unloaded heads", the 256-row bucket): a compute unit pulls ~26 GB/s from HBM however idle the chip is, and a head's cache rows
// (2 x 256 bytes per key) come through ONE unit -- at 250 keys 128 KB, as much as the unit's share of the weights, and the real step
// (cold rows; the timing loop's repeated launches find them in the cache) paid ~4 us per layer for them behind the last q | k | v row.
// With UL the heads' workgroups take NO weight rows: they request their cache rows at the first instruction and poll the granules
// while the other gk - group workgroups of the set stream all (group + 2) * HD rows (RW = 7 instead of 6 for Llama-2-7B).
template <int DT, int NJ, int RW, int NORM, int HD, bool Q8, int PB, bool KT, int NP, bool UL = false>
__global__ void __launch_bounds__(QA_THREADS) k_dec_qkv_attn(const half_t *px, const half_t *pnw, const half_t *pnb, int pcols, int pgeo,
                                                             const uint8_t *pwq, const uint8_t *pwk, const uint8_t *pwv,
                                                             const DecGemvParams P, const DecAttnParams A, const DecQkvAttnExtra E)
{
    // pgeo / pwq / pwk / pwv (round 5): heads | kv_heads << 8 | gk << 16 and the three matrices as leading scalar arguments -- with
    // px / pnw / pnb / pcols they fill the 14 dwords the hardware preloads into SGPRs at wave launch, so that every q | k | v row
    // address is a function of preloaded scalars and the workgroup's id: the weight stream's first request does not wait for
    // the (cold) argument block (k_dec_gemv does the same: A / B on one box 808 vs 798-802 tok/s, profiles/r05_preload_ab.log)
    const int g_heads = pgeo & 0xFF, g_kvh = (pgeo >> 8) & 0xFF, g_gk = (pgeo >> 16) & 0xFFFF;
    const int g_nblk = pcols / block_capacity(DT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long t_kernel = wall_clock64();
    static_assert(NORM == 0 || NORM == 1, "QKV prologue: quantiser with or without the RMS norm");
    static_assert(NP > 0 && NP * 64 < QA_THREADS, "wave-specialised prologue");
    constexpr int TH = QA_THREADS, PT = NP * 64;
    constexpr int MAXC = (NJ * 8 * block_capacity(DT) + PT - 1) / PT;
    XPre<NORM, MAXC, false, PT> pre;
    if (threadIdx.x < PT) pre.issue(px, pnw, pnb, pcols);
    const XLds L = xlds_carve(smem, pcols);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    using Fmt = DecFmt<DT, NJ>;
    const size_t row_bytes = tiled_row_bytes(DT, (size_t)g_nblk);
    // kv group of this workgroup and its rows: local row lr = i * (gk * 8) + (workgroup in group) * 8 + wave, i < RW
    const int g = (int)blockIdx.x / g_gk, bg = (int)blockIdx.x - g * g_gk;
    const int group = g_heads / g_kvh;
    const int RG = (group + 2) * HD;
    // the attention tail: workgroup bg == a * (gk / group) of the set runs query head g * group + a
    const int per = g_gk / group;
    const bool attn_wg = bg % per == 0;
    const int head = g * group + bg / per;
    // waves that take rows, and this wave's place among them (UL: the heads' workgroups are left out)
    const int WGV = UL ? (g_gk - group) * (TH / 64) : g_gk * (TH / 64);
    const int lw = UL ? (bg - bg / per - 1) * (TH / 64) + wave : bg * (TH / 64) + wave;
    struct Row { const uint8_t *w; const half_t *b; half_t *y; int row, vrow; };
    // (selects over kernel-argument scalars, like dec_locate: an indexed read of P.W0[] would be a vector load with a vmcnt(0) behind it)
    auto locate = [&](int lr) {
        Row r;
        const int nq = group * HD;
        const bool isk = lr >= nq && lr < nq + HD, isv = lr >= nq + HD;
        r.row = isv ? g * HD + (lr - nq - HD) : (isk ? g * HD + (lr - nq) : g * nq + lr);
        r.w = isv ? P.W0[2] : (isk ? P.W0[1] : P.W0[0]);
        r.b = isv ? P.b0[2] : (isk ? P.b0[1] : P.b0[0]);
        r.y = isv ? P.y[2] : (isk ? P.y[1] : P.y[0]);
        r.vrow = isv ? (g_heads + g_kvh) * HD + r.row : (isk ? g_heads * HD + r.row : r.row);
        return r;
    };
    typename Fmt::W w[RW];
    // the address of local row lr's weights from the PRELOADED pointers alone (no wait for the argument block)
    auto wrow = [&](int lr) -> const uint8_t * {
        const int nq = group * HD;
        const bool isk = lr >= nq && lr < nq + HD, isv = lr >= nq + HD;
        const int row = isv ? g * HD + (lr - nq - HD) : (isk ? g * HD + (lr - nq) : g * nq + lr);
        // (sums, not a three-way choice of pointers: the compiler turns that choice into a table in scratch memory)
        const uint64_t q = (uint64_t)pwq, dk = (uint64_t)pwk - q, dv = (uint64_t)pwv - q;
        return (const uint8_t *)(q + (isk ? dk : 0) + (isv ? dv : 0)) + (size_t)row * row_bytes;
    };
    auto load_rows = [&](int i0, int i1) {
        const bool full = (RW - 1) * WGV + lw < RG;
        if (full) {
#pragma unroll
            for (int i = 0; i < RW; i++) { if (i < i0 || i >= i1) continue; w[i].load(wrow(i * WGV + lw), g_nblk, lane); }
        } else {
#pragma unroll
            for (int i = 0; i < RW; i++) {
                if (i < i0 || i >= i1) continue;
                if (i > 0 && i * WGV + lw >= RG) continue;
                w[i].load(wrow(min(i * WGV + lw, RG - 1)), g_nblk, lane);
            }
        }
    };
    // the attention tail: workgroup bg == a * (gk / group) of the set runs query head g * group + a, on its waves 0-3 (the
    // prologue waves), which request the head's K / V rows at the entry of dec_attn_body: behind the workgroup's last row in the
    // default mapping (the rows arrive while the granules are polled), at the kernel's first instructions with UL
    static_assert(NP == 4, "the prologue waves are the attention waves");
    const bool ul_head = UL && attn_wg;
    if (!ul_head && threadIdx.x == TH - 1) { L.part[130] = 0.0f; L.part[131] = 0.0f; }
    if (!ul_head && wave >= NP) load_rows(0, 1);          // (before the barrier: see k_dec_gemv)
    // the position and the step's tag: scalar loads through pointers of the argument block -- read BEHIND the first weight
    // requests, which need nothing but preloaded scalars
    const int pos = *(const __attribute__((address_space(4))) int *)(A.state + 1);
    // tag of this step's granules: (decode call, position) -- every granule is rewritten every step, so a tag only has to differ
    // from the previous step's (next position of the same call, or another call)
    const unsigned epoch = ((*(const __attribute__((address_space(4))) unsigned *)(E.epoch) + E.epoch_add) << 20) | ((unsigned)pos & 0xFFFFFu);
    DecAttnFusedIn F;
    F.gran = E.gran; F.epoch = epoch; F.pos = pos; F.err = E.err; F.timeout_ticks = (long long)E.timeout_us * 100;
    F.att_gran = nullptr;
    if constexpr (UL) {
        if (attn_wg) {      // no rows, no prologue, no barrier with the other waves: the cache rows are requested NOW
            if (wave >= 4) {
                // The four waves the attention does not use pull the head's cache rows PAST the entry bucket towards this unit's L2
                // (one dword per 64 bytes of every K / V slice, values unused): the tail's loop asks for those rows batch by batch
                // behind the last q | k | v row, and each batch was a round trip to HBM through one compute unit
#if IFA_QA_WARM_ROWS
                if (pos >= PB) {
                    constexpr int SB = Q8 ? (HD / 32) * 34 : HD * 2;
                    const int kvh = head / group, kv_dim = g_kvh * HD;
                    const size_t rb = Q8 ? (size_t)(kv_dim / 32) * 34 : (size_t)kv_dim * 2;
                    const size_t ho = Q8 ? (size_t)((kvh * HD) / 32) * 34 : (size_t)kvh * HD * 2;
                    const int t = (int)threadIdx.x - 256;
                    const int off = min((t & 3) * 64, SB - 4) & ~1;
                    typedef uint32_t u32_a2 __attribute__((aligned(2)));
                    uint32_t acc = 0;
                    for (int j = PB + (t >> 2); j <= pos; j += 64) {
                        acc ^= *(const __attribute__((address_space(1))) u32_a2 *)(A.kcache + (size_t)j * rb + ho + off);
                        acc ^= *(const __attribute__((address_space(1))) u32_a2 *)(A.vcache + (size_t)j * rb + ho + off);
                    }
                    if (acc == 0x9E3779B9u && A.trace) A.trace[(size_t)A.heads * 8 + 7] = (long long)acc;     // (keeps the loads; never read)
                }
#endif
                return;
            }
            DecAttnRegs<HD, Q8, PB, KT> RU;
            dec_attn_body<HD, Q8, false, PB, KT, true>(smem, nullptr, A.kcache, A.vcache, A.heads, A.kv_heads, A, head, F, RU);
            return;
        }
    }
    __syncthreads();
    DecAttnRegs<HD, Q8, PB, KT> R;
    if (wave >= NP) {
        load_rows(1, RW);
    } else {
        pre.finish(P.norm_w, P.norm_b, P.multi_base, P.eps, P.cols, L, P.xn_out, nullptr);
        load_rows(0, RW);
#if IFA_QA_EARLY_KV
        if (attn_wg) dec_attn_body<HD, Q8, false, PB, KT, true, 1>(smem, nullptr, A.kcache, A.vcache, A.heads, A.kv_heads, A, head, F, R);
#endif
    }
    lds_counter_wait(L.part + 131, NP);
    typename Fmt::X X;
    X.load(L.codes, L.scale, L.xsum, lane, g_nblk);
    {
        float a[RW];
#pragma unroll
        for (int i = 0; i < RW; i++) a[i] = w[i].dot(X);
#pragma unroll
        for (int i = 0; i < RW; i++) a[i] = wave_sum(a[i]);
        float a0 = 0.0f;
#pragma unroll
        for (int i = 0; i < RW; i++) { if (lane == i) a0 = a[i]; }
        const int lr = lane * WGV + lw;
        if (lane < RW && lr < RG) {
            const Row r = locate(lr);
            half_t y = dec_bias(a0, r.b, r.row);
            if (P.pre_scale != 0.0f) y = f2h(h2f(y) * P.pre_scale);
            r.y[r.row] = y;                                     // (the q | k | v buffer stays available to the debug / op paths)
            __hip_atomic_store(E.gran + r.vrow, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(uint16_t, y),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (A.trace && threadIdx.x == 0) {      // tuning: every workgroup's start / rows-published stamps behind the heads' attention stamps
        A.trace[(size_t)(A.heads + (int)blockIdx.x) * 8 + 0] = t_kernel;
        A.trace[(size_t)(A.heads + (int)blockIdx.x) * 8 + 1] = wall_clock64();
    }
    if (attn_wg) {
        __syncthreads();             // every wave of this workgroup is done with the activation image: the LDS is the attention's now
        if (wave >= 4) return;
        dec_attn_body<HD, Q8, false, PB, KT, true, IFA_QA_EARLY_KV ? 2 : 0>(smem, nullptr, A.kcache, A.vcache, A.heads, A.kv_heads, A, head, F, R);
        return;
    }
}

// host side (one translation unit per weight format: ifa_dqkvattn_<format>.hip)
bool dec_qkv_attn_supported(int w_dtype, int cols, int heads, int kv_heads, int head_dim, int num_cus, int *gk_out, int *rw_out);
int dec_qkv_attn_launch(int w_dtype, int norm, bool kv_q8, int pb, bool kt, const DecGemvParams &P, const DecAttnParams &A, const DecQkvAttnExtra &E,
                        int max_ctx, hipStream_t s);
template <int DT>
int dec_qkv_attn_launch_dt(int norm, bool kv_q8, int pb, bool kt, int rw, const DecGemvParams &P, const DecAttnParams &A, const DecQkvAttnExtra &E,
                           int max_ctx, hipStream_t s);

} // namespace ifa
