// ifa_decode_chain.h -- round 6: consecutive GEMV ops of one decoder layer (batch 1) as ONE launch, with the NEXT op's weight
// rows requested BEFORE the hand-off of the current op's output is waited for.
//
//   [Wo (+bias, +residual)] -> [RMS norm -> Q8 quantiser -> W1 | W3 -> activation -> gate] -> [Q8 quantiser -> W2 (+bias, +residual)]
//
// Reference ops: inference_worker.cc:1339-1404 (Wo + Add), :1660-1923 (FFN), kernels gemv.h:1580-1709, tensor_quant.h:44-82,
// with the row / prologue / epilogue code of k_dec_gemv (ifa_decode_kernels.h): the chain is bit-identical to the separate
// launches (tests/test_gpu_chain.py).
//
// Why this and not the r3 / r4 forms.  A hand-off through memory costs ~3.3 us on this part (write-through store, visibility,
// one load round trip behind the CU's own queue) against ~2.5 us for a kernel boundary + first load (DESIGN.md, r4 price list):
// a fused launch only wins through what STREAMS during the hand-off.  The r3 engine streamed into a 112 KiB LDS ring and paid
// for it with issue-bound consumer waves; the r4 fusions split the waves (or the CUs) into a front part and loaders, so the
// front part's round trips queued behind the loaders' requests.  Here every wave is both: the register file is the ring
// (512 KiB per CU; a wave keeps 6-17 KiB of the next op's rows in its own registers, the same registers and the same code
// k_dec_gemv streams through at 7 TB/s), and a wave requests its next rows only after its own stores are out, so nothing of
// the dependency chain sits behind more than one op's worth of its own requests.
//
// Hand-off: one 4-byte granule {tag16 << 16 | half} per output row, written by ONE agent-scope relaxed store (write-through,
// MI355X_MICROARCH.md form R2: the data is the flag) and gathered 16 bytes (4 granules) per request with sc1 loads; a thread
// owns a chunk of 8 consecutive values -- the chunk of XPre's quantiser, so norm statistic and codes are the bits of the
// separate launches -- and re-requests only the chunks whose tags are not this step's yet.  The gather is ONE pass in the
// common case: every workgroup raises a flag (its last publishing wave, through an LDS counter) and ONE wave per CU polls the
// 1 KB of flags while the others wait on an LDS word -- 1024 threads of 256 CUs polling the 44 KB payload itself was 11 MB per
// round, starved the weight stream and took 15-25 us (profiles/r06_chain_first_trace.log).  The flag is not ordered behind
// the other waves' granule stores: it is a hint, the tags decide.  tag16 = 0x8000 | (call & 127) << 8
// | (position & 255): every granule of a layer is rewritten every step, so a tag only has to differ from the previous
// step's (next position of the same call, or another call).  One workgroup per CU, all resident (wait_grid_fits); every wait
// is bounded (error word -> the call fails, the waiting launches go off: ifa_runtime.hip).
#pragma once
#include "ifa_decode_kernels.h"

namespace ifa {

struct DecChainExtra {
    uint32_t *gran_a;            // [dim] this layer's Wo-output granules (WO chains), else unused
    uint32_t *gran_h;            // [ffn] this layer's gated-product granules
    uint32_t *flags_a, *flags_h; // [grid] per workgroup: the tag of the step whose Wo rows / gated rows it has published (a hint: see below)
    const int *state;            // state[1] = position of the step
    const unsigned *epoch;       // device word: decode-call counter
    unsigned epoch_add;          // added to it (ifa_model_time_kernel: distinct tags for repeated launches of one step)
    unsigned *err;               // error word
    int timeout_us;
    long long *trace;            // optional [grid][16] wall-clock stamps (100 MHz)
    int late_w2;                 // measurement only: the loaders request their W2 rows when the image is there (no prefetch: the hand-off on a quiet memory system)
};

__device__ __forceinline__ uint32_t chain_tag(unsigned call, int pos) { return 0x8000u | ((call & 127u) << 8) | ((unsigned)pos & 255u); }

typedef uint32_t ch_u32x4 __attribute__((ext_vector_type(4)));

// Gather of this thread's chunks (chunk c = tid + k * nthr: 8 consecutive values = 8 granules = two 16-byte requests) into the
// registers XPre::finish quantises.  Returns false when the wait gave up (error word written).
template <int MAXC>
__device__ __forceinline__ bool chain_gather(half8_t (&xv)[MAXC], const uint32_t *gran, int cols, int nthr, uint32_t tag,
                                             long long t_give_up, unsigned *err, unsigned code)
{
    const int chunks = cols >> 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(gran), 0, cols * 4, 0x00020000);
    ch_u32x4 lo[MAXC], hi[MAXC];
    unsigned have = 0;
    bool good = true;
    for (;;) {
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            const int c = (int)threadIdx.x + k * nthr;
            if (c < chunks && !((have >> k) & 1u)) {
                lo[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, c * 32, 0, 16);          // aux 16 = sc1: past this CU's L1
                hi[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, c * 32 + 16, 0, 16);
            }
        }
        bool all = true;
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            const int c = (int)threadIdx.x + k * nthr;
            if (c < chunks && !((have >> k) & 1u)) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 4; i++) ok = ok && (lo[k][i] >> 16) == tag && (hi[k][i] >> 16) == tag;
                if (ok) have |= 1u << k; else all = false;
            }
        }
        if (all) break;
        if (wall_clock64() > t_give_up) { atomicExch(err, code); good = false; break; }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            xv[k][i] = __builtin_bit_cast(half_t, (uint16_t)lo[k][i]);
            xv[k][4 + i] = __builtin_bit_cast(half_t, (uint16_t)hi[k][i]);
        }
    }
    return good;
}

__device__ __forceinline__ void chain_publish(uint32_t *gran, int row, uint32_t tag, half_t y)
{
    __hip_atomic_store(gran + row, (tag << 16) | (uint32_t)__builtin_bit_cast(uint16_t, y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ONE wave polls the workgroups' flags (grid <= 1024 dwords: 16 bytes per lane and round) until every one carries this step's tag
__device__ __forceinline__ bool chain_poll_flags(const uint32_t *flags, int grid, uint32_t tag, int lane, long long t_give_up, unsigned *err, unsigned code)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(flags), 0, grid * 4, 0x00020000);
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i0 = (lane + 64 * r) * 4;
            if (i0 < grid) {
                const ch_u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(rs, i0 * 4, 0, 16);
#pragma unroll
                for (int j = 0; j < 4; j++) ok = ok && (i0 + j >= grid || f[j] == tag);
            }
        }
        if (__all(ok)) return true;
        if (wall_clock64() > t_give_up) { if (lane == 0) atomicExch(err, code); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
}

// LDS control words of a chained launch (behind the activation images)
enum { CH_C_PUB_A = 0, CH_C_PUB_H = 1, CH_C_GO_A = 2, CH_C_GO_H = 3, CH_C_WORDS = 4 };
// a wave has published its rows of an op: the workgroup's last one raises the flag
__device__ __forceinline__ void chain_wave_published(float *ctl_word, int nwaves, uint32_t *flag, uint32_t tag, int lane, long long *stamp = nullptr)
{
    if (lane == 0) {
        asm volatile("" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(ctl_word), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int)old == nwaves - 1) {
            __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (stamp) *stamp = wall_clock64();
        }
    }
}
__device__ __forceinline__ void chain_go(float *ctl_word)
{
    __hip_atomic_store(reinterpret_cast<uint32_t *>(ctl_word), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

constexpr int CH_TRACE = 16;

// DT: weight format of all matrices.  NJA / NJB: blocks per lane of a dim-wide / ffn-wide row.  RW: W1 (| W3) rows (pairs) per wave
// and pass.  EPI: EPI_GLU or EPI_ACT.  NORM: 1 = RMS norm in front of the FFN quantiser, 0 = none.  WO: the Wo rows in front
// (their input: the XqImage the attention kernel left, px).  RO: Wo rows per wave (all waves).  R2: W2 rows per LOADER wave.
// TH: threads.  Leading scalars as in k_dec_gemv (preloaded into SGPRs): px = the FFN input (F16) or, WO, the attention image.
//
// Wave roles.  Waves [0, NP) are the workgroup's "front" waves, waves [NP, NW) its "loaders" (NP = NW / 2, k_dec_gemv's split):
//   * a hand-off's round trips (flag poll, gather) are made by the front waves, which at that moment have NO weight request of
//     their own in flight -- a wave's loads return in order, so a gather issued behind the wave's own rows would only be usable
//     once those have arrived (first form of this kernel: 9-13 us per hand-off, profiles/r06_chain_first_trace.log);
//   * the loaders request the next op's rows as soon as their own stores are out and wait on an LDS word;
//   * W1 | W3: front waves request their rows after the quantiser (they finish last, which makes them the waves with nothing
//     in flight at the next hand-off); W2: only the loaders take rows (R2 each), the front waves gather, quantise and exit.
template <int DT, int NJA, int NJB, int RW, int EPI, int NORM, bool WO, int RO, int R2, int TH>
__global__ void __launch_bounds__(TH) k_dec_chain(const half_t *px, const half_t *pnw, const half_t *pnb, int pcols,
                                                  const uint8_t *pw0, const uint8_t *pw1, int pnblk_grid, int ptotal,
                                                  const DecGemvParams P, const DecGemvParams Q, const DecGemvParams PW, const DecChainExtra E)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(EPI == EPI_GLU || EPI == EPI_ACT, "FFN up-projection epilogues");
    static_assert(NORM == 0 || NORM == 1, "FFN prologue");
    const long long t_kernel = wall_clock64();
    constexpr int NW = TH / 64, NP = NW / 2, PT = NP * 64, NL = NW - NP;
    constexpr int NM = EPI == EPI_GLU ? 2 : 1;
    constexpr int MAXC = (NJA * 8 * block_capacity(DT) + PT - 1) / PT;
    constexpr int MAXC2 = (NJB * 8 * block_capacity(DT) + PT - 1) / PT;
    using FmtA = DecFmt<DT, NJA>;
    using Fmt1 = DecFmt<DT, 1>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * NW + wave;
    const int G = (int)((unsigned)pnblk_grid >> 16);
    const int W = G * NW;
    const int nblk_a = pnblk_grid & 0xFFFF, rows13 = ptotal;
    const size_t row_bytes_a = tiled_row_bytes(DT, (size_t)nblk_a);
    const int npass = (rows13 + RW * W - 1) / (RW * W);
    // LDS: [image of the dim-wide input][image of the ffn-wide input][control words] -- two regions on purpose: a thread leaves
    // the gather as soon as ITS chunks are there and writes its part of the second image while slower waves of the workgroup
    // may still be reading the first
    const XLds L = xlds_carve(smem, pcols);
    const size_t off2 = (xlds_bytes(pcols) + 15) / 16 * 16;
    const XLds L2 = xlds_carve(smem + off2, Q.cols);
    float *const ctl = reinterpret_cast<float *>(smem + off2 + (xlds_bytes(Q.cols) + 15) / 16 * 16);

    XPre<NORM, MAXC, false, PT> pre;
    typename FmtA::X XA;
    typename FmtA::W w[NM][RW];
    auto load_rows13 = [&](int pass, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < RW; i++) {
            if (i < i0 || i >= i1) continue;
            const int v = (pass * RW + i) * W + gw;
            if (i > 0 && v >= rows13) continue;                  // (rows past the end are not requested; row 0 of a pass is clamped)
            const int vc = min(v, rows13 - 1);
            w[0][i].load(pw0 + (size_t)vc * row_bytes_a, nblk_a, lane);
            if constexpr (NM == 2) w[1][i].load(pw1 + (size_t)vc * row_bytes_a, nblk_a, lane);
        }
    };
    long long *const trc = (E.trace != nullptr && (threadIdx.x == 64 || threadIdx.x == TH - 64)) ? E.trace + (size_t)blockIdx.x * CH_TRACE + (threadIdx.x == 64 ? 0 : 10) : nullptr;
    const bool trf = trc != nullptr && threadIdx.x == 64;        // stamps of a front wave [0, 10), of a loader wave [10, 16)
    const bool trl = trc != nullptr && threadIdx.x != 64;
    uint32_t tag = 0;
    long long t_give_up = 0;
    auto read_tag = [&]() {
        const int pos = *(const __attribute__((address_space(4))) int *)(E.state + 1);
        tag = chain_tag(*(const __attribute__((address_space(4))) unsigned *)(E.epoch) + E.epoch_add, pos);
        t_give_up = t_kernel + (long long)E.timeout_us * 100;
    };
    auto zero_ctl = [&]() {
        if (threadIdx.x == TH - 1) {
            L.part[130] = 0.0f; L.part[131] = 0.0f; L2.part[130] = 0.0f; L2.part[131] = 0.0f;
#pragma unroll
            for (int i = 0; i < CH_C_WORDS; i++) ctl[i] = 0.0f;
        }
    };

#ifndef IFA_CHAIN_NO_PRIO
    if (wave < NP) __builtin_amdgcn_s_setprio(3);      // the hand-offs' round trips are issued by the front waves: ahead of the loaders' requests
#endif
    if constexpr (WO) {
        // ---- Wo rows on the quantised attention output (k_dec_gemv<EPI_RESIDUAL, NORM 2>): first in every wave's queue, the
        // FFN stream is requested behind the published rows
        if (wave < NP) pre.issue_norm(pnw, pnb, pcols);
        const XqImage img = xq_image_carve(const_cast<half_t *>(px), pcols);
        XA.load(img.codes, img.scale, img.xsum, lane, nblk_a);
        typename FmtA::W wo[RO];
        const int rows_o = PW.total_rows;
#pragma unroll
        for (int i = 0; i < RO; i++) {
            const int v = i * W + gw;
            if (i > 0 && v >= rows_o) continue;
            wo[i].load(PW.W0[0] + (size_t)min(v, rows_o - 1) * row_bytes_a, nblk_a, lane);
        }
        const half_t res = PW.residual[min(min(lane, RO - 1) * W + gw, rows_o - 1)];
        zero_ctl();
        read_tag();
        if (trf) { trc[0] = t_kernel; trc[1] = wall_clock64(); }
        __syncthreads();                 // (the LDS counters are zero before any wave counts)
        float aw[RO];
#pragma unroll
        for (int i = 0; i < RO; i++) aw[i] = (i == 0 || i * W + gw < rows_o) ? wo[i].dot(XA) : 0.0f;
#pragma unroll
        for (int i = 0; i < RO; i++) aw[i] = wave_sum(aw[i]);
        float a0 = 0.0f;
#pragma unroll
        for (int i = 0; i < RO; i++) { if (lane == i) a0 = aw[i]; }
        const int v = lane * W + gw;
        if (lane < RO && v < rows_o) {
            const DecRow d = dec_locate(PW, v);
            const half_t y = dec_row_value<EPI_RESIDUAL>(PW, d, a0, 0.0f, res, (half_t)0);
            chain_publish(E.gran_a, d.row, tag, y);
            d.y[d.row] = y;              // (the plain copy: debug surface, op path)
        }
        chain_wave_published(ctl + CH_C_PUB_A, NW, E.flags_a + blockIdx.x, tag, lane);
        if (trf) trc[2] = wall_clock64();
        if (trl) trc[0] = wall_clock64();
        if (wave >= NP) {
            load_rows13(0, 0, RW);
        } else {
            if (wave == 0) { chain_poll_flags(E.flags_a, G, tag, lane, t_give_up, E.err, 0x93u); chain_go(ctl + CH_C_GO_A); }
            else lds_counter_wait(ctl + CH_C_GO_A, 1);
            chain_gather<MAXC>(pre.xv, E.gran_a, pcols, PT, tag, t_give_up, E.err, 0x91u);
            if (trf) trc[3] = wall_clock64();
            pre.finish(P.norm_w, P.norm_b, P.multi_base, P.eps, pcols, L, nullptr, nullptr);
            load_rows13(0, 0, RW);
        }
        lds_counter_wait(L.part + 131, NP);
    } else {
        // ---- the FFN input is in memory when the launch starts: k_dec_gemv's wave-specialised prologue
        if (threadIdx.x < PT) pre.issue(px, pnw, pnb, pcols);
        zero_ctl();
        if (NORM == 1 && wave >= NP) load_rows13(0, 0, 1);
        read_tag();
        if (trf) { trc[0] = t_kernel; trc[1] = wall_clock64(); }
        __syncthreads();
        if (wave >= NP) {
            load_rows13(0, NORM == 1 ? 1 : 0, RW);
        } else {
            pre.finish(P.norm_w, P.norm_b, P.multi_base, P.eps, pcols, L, nullptr, nullptr);
            load_rows13(0, 0, RW);
        }
        if (trf) trc[2] = trc[3] = wall_clock64();
        if (trl) trc[0] = wall_clock64();
        lds_counter_wait(L.part + 131, NP);
    }
    XA.load(L.codes, L.scale, L.xsum, lane, nblk_a);
    if (trf) trc[4] = wall_clock64();
    if (trl) trc[1] = wall_clock64();

    // ---- W1 | W3 rows, activation, gate: every finished row is a granule
    for (int pass = 0; pass < npass; pass++) {
        if (pass > 0) load_rows13(pass, 0, RW);
        float a[NM][RW];
#pragma unroll
        for (int i = 0; i < RW; i++)
#pragma unroll
            for (int m = 0; m < NM; m++) a[m][i] = ((pass * RW + i) * W + gw < rows13 || i == 0) ? w[m][i].dot(XA) : 0.0f;
#pragma unroll
        for (int i = 0; i < RW; i++)
#pragma unroll
            for (int m = 0; m < NM; m++) a[m][i] = wave_sum(a[m][i]);
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int i = 0; i < RW; i++) {
            if (lane == i) { a0 = a[0][i]; if constexpr (NM == 2) a1 = a[1][i]; }
        }
        const int v = (pass * RW + lane) * W + gw;
        if (lane < RW && v < rows13) {
            DecRow d; d.si = 0; d.row = v; d.W0 = nullptr; d.W1 = nullptr; d.b0 = P.b0[0]; d.b1 = P.b1; d.y = P.y[0];
            const half_t y = dec_row_value<EPI>(P, d, a0, a1);
            chain_publish(E.gran_h, v, tag, y);
            d.y[v] = y;
        }
    }
    chain_wave_published(ctl + CH_C_PUB_H, NW, E.flags_h + blockIdx.x, tag, lane, E.trace ? E.trace + (size_t)blockIdx.x * CH_TRACE + 9 : nullptr);
    if (trf) trc[5] = wall_clock64();
    if (trl) trc[2] = wall_clock64();

    // ---- W2
    const int nblk_b = Q.nblk, rows2 = Q.total_rows;
    if (wave < NP) {
        // front waves: nothing of their own in flight -- flags, gather, quantiser, image; then they are done
        XPre<0, MAXC2, false, PT> pre2;
        if (wave == 0) { chain_poll_flags(E.flags_h, G, tag, lane, t_give_up, E.err, 0x94u); chain_go(ctl + CH_C_GO_H); }
        else lds_counter_wait(ctl + CH_C_GO_H, 1);
        if (trf) trc[6] = wall_clock64();
        chain_gather<MAXC2>(pre2.xv, E.gran_h, Q.cols, PT, tag, t_give_up, E.err, 0x92u);
        if (trf) trc[7] = wall_clock64();
        pre2.finish(nullptr, nullptr, 0.0f, Q.eps, Q.cols, L2, nullptr, nullptr);
        if (trf) trc[8] = wall_clock64();
        return;
    }
    // loaders: R2 rows each, requested now (this wave's stores are out), one 64-block slice per register set so that the
    // activation slice can be read from LDS per step (two whole rows + the whole image would not fit 128 registers)
    // ... but not before every wave of THIS workgroup has its gated rows out: the CU's memory queue has no priorities, and W2
    // requests of its early waves delayed the W1 | W3 requests its late waves still had to get accepted (last gated row
    // of the chip at 16.8 us instead of 10.7: profiles/r06_chain_trace_roles.log)
#ifndef IFA_CHAIN_NO_GATE
    lds_counter_wait(ctl + CH_C_PUB_H, NW);
#endif
    if (E.late_w2) lds_counter_wait(L2.part + 131, NP);
    const size_t row_bytes_b = tiled_row_bytes(DT, (size_t)nblk_b);
    const int lw = blockIdx.x * NL + (wave - NP), WL = G * NL;
    typename Fmt1::W w2[R2][NJB];
#pragma unroll
    for (int i = 0; i < R2; i++) {
        const int v = i * WL + lw;
        if (i > 0 && v >= rows2) continue;
        const uint8_t *wr = Q.W0[0] + (size_t)min(v, rows2 - 1) * row_bytes_b;
#pragma unroll
        for (int j = 0; j < NJB; j++) w2[i][j].load(wr, nblk_b, lane, 64 * j);
    }
    half_t res = (half_t)0, res2 = (half_t)0;
    {
        const int row = min(min(lane, R2 - 1) * WL + lw, rows2 - 1);
        // WO: the residual is a row another wave published in THIS launch -- read it the way the gather does (its plain copy
        // is not visible to this CU before the launch ends); every Wo row is out long before the gated rows are
        if constexpr (WO) res = __builtin_bit_cast(half_t, (uint16_t)__hip_atomic_load(E.gran_a + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        else res = Q.residual[row];
        if (Q.residual2) res2 = Q.residual2[row];
    }
    if (trl) trc[3] = wall_clock64();
    lds_counter_wait(L2.part + 131, NP);
    if (trl) trc[4] = wall_clock64();
    if (lw >= rows2) return;
    float a2[R2];
#pragma unroll
    for (int i = 0; i < R2; i++) a2[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < NJB; j++) {
        typename Fmt1::X Xj;
        Xj.load(L2.codes, L2.scale, L2.xsum, lane, nblk_b, 64 * j);
#pragma unroll
        for (int i = 0; i < R2; i++) a2[i] = (i == 0 || i * WL + lw < rows2) ? w2[i][j].dot(Xj, a2[i]) : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < R2; i++) a2[i] = wave_sum(a2[i]);
    float b0 = 0.0f;
#pragma unroll
    for (int i = 0; i < R2; i++) { if (lane == i) b0 = a2[i]; }
    const int v2 = lane * WL + lw;
    if (lane < R2 && v2 < rows2) dec_finish_row<EPI_RESIDUAL>(Q, dec_locate(Q, v2), b0, 0.0f, res, res2);
    if (trl) trc[5] = wall_clock64();
}

// host side (one translation unit per weight format: ifa_dchain_<format>.hip)
// wo: the Wo rows ride in front (PW: launch_wo's EPI_RESIDUAL / NORM 2 parameters).  P: launch_ffn13's dense parameters (EPI_GLU /
// EPI_ACT).  Q: launch_w2's EPI_RESIDUAL parameters.
bool dec_chain_supported(int w_dtype, int w2_dtype, int wo_dtype, int dim, int ffn, bool wo, int wo_cols, int num_cus);
int dec_chain_launch(int w_dtype, bool glu, int norm, bool wo, const DecGemvParams &P, const DecGemvParams &Q, const DecGemvParams *PW,
                     const DecChainExtra &E, int num_cus, hipStream_t s);
template <int DT>
int dec_chain_launch_dt(bool glu, int norm, bool wo, const DecGemvParams &P, const DecGemvParams &Q, const DecGemvParams &PW,
                        const DecChainExtra &E, int num_cus, hipStream_t s);

} // namespace ifa
