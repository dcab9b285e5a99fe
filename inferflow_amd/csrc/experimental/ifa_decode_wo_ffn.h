// ifa_decode_wo_ffn.h -- round 4: the Wo rows inside the W1 / W3 launch (one launch instead of two), split BY WORKGROUP.
//
// Reference ops (batch 1): Quantize(attention output) -> Wo GEMV -> +bias -> Add(residual)           (inference_worker.cc:1339-1404)
//                          -> RmsNorm -> Quantize -> W1 GEMV, W3 GEMV -> activation -> Mul             (:1660-1923)
// with the arithmetic of k_dec_gemv<EPI_RESIDUAL, NORM 2> and k_dec_gemv<EPI_GLU, NORM 1> (shared row / prologue / epilogue code:
// fused == separate launches bit for bit, tests/test_gpu_fused_attn.py).
//
// The first form of this kernel split the work by WAVE (front waves: Wo + all-gather + quantiser; loader waves: the W1 / W3
// stream) and was slower than two launches: every store, flag and poll of the front waves queued behind the loader waves'
// requests in the SAME compute unit's memory pipeline (1.2-1.5 us per round trip, 4.4 us until the Wo rows were out, 5-6 us for
// the gather: profiles/r04_fused_launch_phase_trace.log).  Here the split is by COMPUTE UNIT:
//   * 64 FRONT workgroups (8 per XCD) compute the Wo rows -- 64 rows each, all requested at once -- store them write-through,
//     exchange them among themselves (64 flags, queues empty: sub-microsecond round trips), run the FFN norm + quantiser, and
//     publish the QUANTISED image (5 KB: codes | scales | sums, each front workgroup its 2 blocks) + a flag; then they are done;
//   * 192 LOADER workgroups request ALL their W1 / W3 rows at once and then ONE poll of the 64 image flags: it sits behind their
//     own weight requests and comes back when those have drained -- which is when the image is needed; they copy the image into
//     LDS (past the caches), and finish their rows.  No prologue on their critical path at all.
// The whole Wo launch -- boundary, 1 us to its first request, 1.2 us to its first byte -- and the W1 / W3 prologue disappear
// under the weight stream.  Tags = (decode call, position) as in ifa_decode_qkv_attn.h; bounded waits, error word.
#pragma once
#include "ifa_decode_kernels.h"

namespace ifa {

struct DecWoFfnExtra {
    unsigned long long *a_flags;    // [nfront] {tag, 1}: front workgroup f's Wo rows are in memory
    unsigned long long *img_flags;  // [nfront] {tag, 1}: its blocks of the quantised FFN input are in memory
    void *img;                      // the quantised FFN input, XqImage layout (codes | scales | sums) of `dim` columns
    const int *state;               // state[1] = position of the step (tag)
    const unsigned *epoch;          // device word: decode-call counter (tag)
    unsigned epoch_add;
    unsigned *err;
    int timeout_us;
    long long *trace;               // optional [grid][8] wall-clock stamps (100 MHz): see tools/trace_fused.py
};

constexpr int WF_THREADS = 1024, WF_FRONT = 64;      // front workgroups of a 256-workgroup grid: ids whose (id / 8) % 4 == 0 (8 per XCD)

// RW: row pairs of a loader wave (rows [0, RW * loader waves), strided); RWF: of a front wave (the rest, requested once its image
// is in LDS: few); RWO: Wo rows of a front wave (rows = WF_FRONT * 16 * RWO)
template <int DT, int NJ, int RW, int RWF, int RWO, int EPI>
__global__ void __launch_bounds__(WF_THREADS) k_dec_wo_ffn(const half_t *pxq, const half_t *pnw, const half_t *pnb, int pcols,
                                                           const DecGemvParams PW, const DecGemvParams P, const DecWoFfnExtra E)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long t_kernel = wall_clock64();
    static_assert(EPI == EPI_GLU || EPI == EPI_ACT, "FFN up-projection epilogues");
    constexpr int TH = WF_THREADS, NW = TH / 64, NPRO = 8, PT = NPRO * 64;      // the quantiser runs on 8 of a front workgroup's 16 waves
    constexpr int NM = EPI == EPI_GLU ? 2 : 1;
    constexpr int MAXC = (NJ * 8 * block_capacity(DT) + PT - 1) / PT;
    using Fmt = DecFmt<DT, NJ>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunks = pcols >> 3;
    const int fstep = ((int)gridDim.x >> 3) / (WF_FRONT >> 3);                  // every fstep-th group of 8 workgroups is a front group
    const int grp = (int)blockIdx.x >> 3;
    const bool front = grp % fstep == 0;
    const int f = (grp / fstep) * 8 + ((int)blockIdx.x & 7);                    // front index
    const int lidx = (int)blockIdx.x - 8 * ((grp + fstep - 1) / fstep);         // loader index (front groups with a smaller id removed)
    const int NL = (int)gridDim.x - WF_FRONT;
    long long *const trc = (E.trace && threadIdx.x == 0) ? E.trace + (size_t)blockIdx.x * 8 : nullptr;
    if (trc) { trc[0] = t_kernel; trc[7] = front ? 1 : 0; }
    const int pos = *(const __attribute__((address_space(4))) int *)(E.state + 1);
    const unsigned epoch = ((*(const __attribute__((address_space(4))) unsigned *)(E.epoch) + E.epoch_add) << 20) | ((unsigned)pos & 0xFFFFFu);
    const unsigned long long tagw = (unsigned long long)epoch << 32;
    const XLds L = xlds_carve(smem, pcols);
    const long long t_give_up = t_kernel + (long long)E.timeout_us * 100;
    if (threadIdx.x == TH - 1) { L.part[128] = 0.0f; L.part[129] = 0.0f; L.part[130] = 0.0f; L.part[131] = 0.0f; }

    if (front) {
        // ================================================================= front workgroup: Wo rows [f * 16 * RWO, + 16 * RWO)
        XPre<1, MAXC, false, PT> pre;
        typename Fmt::X XW;
        typename Fmt::W wo[RWO];
        const size_t wo_row_bytes = tiled_row_bytes(DT, (size_t)PW.nblk);
        const int row0 = (f * NW + wave) * RWO;
        const XqImage Q = xq_image_carve(const_cast<half_t *>(pxq), PW.cols);
        XW.load(Q.codes, Q.scale, Q.xsum, lane, PW.nblk);
#pragma unroll
        for (int i = 0; i < RWO; i++) wo[i].load(PW.W0[0] + (size_t)min(row0 + i, PW.total_rows - 1) * wo_row_bytes, PW.nblk, lane);
        const half_t wres = PW.residual[min(row0 + min(lane, RWO - 1), PW.total_rows - 1)];
        if (wave < NPRO) {
#pragma unroll
            for (int k = 0; k < MAXC; k++) {
                const int c = (int)threadIdx.x + k * PT;
                if (c < chunks) {
                    if (pnw) pre.wv[k] = *reinterpret_cast<const half8_t *>(pnw + (size_t)c * 8);
                    if (pnb) pre.bv[k] = *reinterpret_cast<const half8_t *>(pnb + (size_t)c * 8);
                }
            }
        }
        __syncthreads();                                  // (the LDS counters are zero)
        float aw[RWO];
#pragma unroll
        for (int i = 0; i < RWO; i++) aw[i] = wo[i].dot(XW);
#pragma unroll
        for (int i = 0; i < RWO; i++) aw[i] = wave_sum(aw[i]);
        float a0 = 0.0f;
#pragma unroll
        for (int i = 0; i < RWO; i++) { if (lane == i) a0 = aw[i]; }
        if (lane < RWO && row0 + lane < PW.total_rows) {
            half_t y = dec_bias(a0, PW.b0[0], row0 + lane);
            y = f2h(h2f(wres) + h2f(y));                  // TensorOpr::Add (half)
            __hip_atomic_store(reinterpret_cast<uint16_t *>(PW.y[0]) + row0 + lane, __builtin_bit_cast(uint16_t, y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's rows are in memory
        if (lane == 0) lds_counter_add(L.part + 129);
        if (wave == 0) {
            lds_counter_wait(L.part + 129, NW);
            if (lane == 0) __hip_atomic_store(E.a_flags + f, tagw | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (trc) trc[1] = wall_clock64();
            for (;;) {                                     // the other front workgroups' rows
                const bool ok = lane >= WF_FRONT || (unsigned)(__hip_atomic_load(E.a_flags + min(lane, WF_FRONT - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == epoch;
                if (__all(ok)) break;
                if (wall_clock64() > t_give_up) { if (lane == 0) atomicExch(E.err, 0x61u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) lds_counter_add(L.part + 128);
        }
        lds_counter_wait(L.part + 128, 1);
        if (trc) trc[2] = wall_clock64();
        if (wave < NPRO) {
            // the whole vector a, past the caches (an older copy of the buffer may sit in this XCD's L2), then the standard norm +
            // quantiser of the FFN input among these 8 waves
#pragma unroll
            for (int k = 0; k < MAXC; k++) {
                const int c = (int)threadIdx.x + k * PT;
                if (c >= chunks) continue;
                const unsigned long long *g = reinterpret_cast<const unsigned long long *>(PW.y[0] + (size_t)c * 8);
                const unsigned long long lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 both = {lo, hi};
                pre.xv[k] = __builtin_bit_cast(half8_t, both);
            }
            pre.finish(P.norm_w, P.norm_b, P.multi_base, P.eps, P.cols, L, P.xn_out, nullptr);
        }
        if (wave == 0) {
            lds_counter_wait(L.part + 131, NPRO);
            if (trc) trc[3] = wall_clock64();
            // publish this workgroup's slice of the image: the blocks of columns [f * cols / WF_FRONT, + cols / WF_FRONT)
            const XqImage G = xq_image_carve(E.img, P.cols);
            const int cper = P.cols / WF_FRONT, c0 = f * cper;             // columns of the slice (a multiple of 32)
            for (int i = lane; i < cper / 4; i += 64)
                __hip_atomic_store(reinterpret_cast<uint32_t *>(G.codes + c0) + i, reinterpret_cast<const uint32_t *>(L.codes + c0)[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = lane; i < cper / 32; i += 64) {
                __hip_atomic_store(reinterpret_cast<uint32_t *>(G.scale + c0 / 32) + i, __builtin_bit_cast(uint32_t, L.scale[c0 / 32 + i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<uint32_t *>(G.xsum + c0 / 32) + i, __builtin_bit_cast(uint32_t, L.xsum[c0 / 32 + i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(E.img_flags + f, tagw | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (trc) trc[4] = wall_clock64();
        }
        // ---- the W1 (| W3) row pairs the loaders do not take: [RW * loader waves, rows), strided over the front waves
        lds_counter_wait(L.part + 131, NPRO);
        const int WLd = NL * NW, base = min(RW * WLd, P.total_rows), gf = f * NW + wave, WFr = WF_FRONT * NW;
        if (base + gf >= P.total_rows) return;
        const size_t frow_bytes = tiled_row_bytes(DT, (size_t)P.nblk);
        typename Fmt::W wf[NM][RWF];
#pragma unroll
        for (int i = 0; i < RWF; i++) {
            const int v = min(base + i * WFr + gf, P.total_rows - 1);
            if (i > 0 && base + i * WFr + gf >= P.total_rows) continue;
            wf[0][i].load(P.W0[0] + (size_t)v * frow_bytes, P.nblk, lane);
            if constexpr (NM == 2) wf[1][i].load(P.W1 + (size_t)v * frow_bytes, P.nblk, lane);
        }
        typename Fmt::X XF;
        XF.load(L.codes, L.scale, L.xsum, lane, P.nblk);
        float af[NM][RWF];
#pragma unroll
        for (int i = 0; i < RWF; i++)
#pragma unroll
            for (int m = 0; m < NM; m++) af[m][i] = (i == 0 || base + i * WFr + gf < P.total_rows) ? wf[m][i].dot(XF) : 0.0f;
#pragma unroll
        for (int i = 0; i < RWF; i++)
#pragma unroll
            for (int m = 0; m < NM; m++) af[m][i] = wave_sum(af[m][i]);
        float f0 = 0.0f, f1 = 0.0f;
#pragma unroll
        for (int i = 0; i < RWF; i++) {
            if (lane == i) { f0 = af[0][i]; if constexpr (NM == 2) f1 = af[1][i]; }
        }
        const int vf = base + lane * WFr + gf;
        if (lane < RWF && vf < P.total_rows) dec_finish_row<EPI>(P, dec_locate(P, vf), f0, f1);
        if (trc) trc[5] = wall_clock64();
        return;
    }

    // ===================================================================== loader workgroup: W1 (| W3) row pairs, strided over the loader waves
    const int gw = lidx * NW + wave, W = NL * NW;         // (rows >= RW * W belong to the front workgroups)
    const size_t row_bytes = tiled_row_bytes(DT, (size_t)P.nblk);
    typename Fmt::W w[NM][RW];
    {
        const bool full = (RW - 1) * W + gw < P.total_rows;
        auto one = [&](int i) {
            const int v = min(i * W + gw, P.total_rows - 1);
            w[0][i].load(P.W0[0] + (size_t)v * row_bytes, P.nblk, lane);
            if constexpr (NM == 2) w[1][i].load(P.W1 + (size_t)v * row_bytes, P.nblk, lane);
        };
        // PACED start: the 64 front workgroups have their whole Wo share (164 KB each) queued from the first instruction; requested
        // against 192 loaders with everything queued too, the Wo rows took 8.5 us to arrive (equal shares of the memory system per
        // CU).  The loaders therefore open with ONE row pair on half of their waves (40 KB per CU) and hold the rest back for
        // E.timeout-independent 2.2 us: the Wo stream gets most of the bandwidth while it lasts.
        const bool late = wave >= NW / 2;
        if (late) { while (wall_clock64() - t_kernel < 220) __builtin_amdgcn_s_sleep(4); }
        one(0);
        if (!late) { while (wall_clock64() - t_kernel < 220) __builtin_amdgcn_s_sleep(4); }
        if (full) {
#pragma unroll
            for (int i = 1; i < RW; i++) one(i);
        } else {
#pragma unroll
            for (int i = 1; i < RW; i++) { if (i * W + gw >= P.total_rows) continue; one(i); }
        }
    }
    if (trc) trc[1] = wall_clock64();
    __syncthreads();                                      // (the LDS counters are zero)
    if (wave == 0) {
        // ONE flag sweep behind this workgroup's weight requests: it comes back when they have drained
        for (;;) {
            const bool ok = lane >= WF_FRONT || (unsigned)(__hip_atomic_load(E.img_flags + min(lane, WF_FRONT - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == epoch;
            if (__all(ok)) break;
            if (wall_clock64() > t_give_up) { if (lane == 0) atomicExch(E.err, 0x62u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) lds_counter_add(L.part + 128);
    }
    lds_counter_wait(L.part + 128, 1);
    if (trc) trc[2] = wall_clock64();
    {   // the image into LDS, 8 bytes per thread, past the caches (XqImage and XLds share the layout: codes | scales | sums)
        const int words = (P.cols + (P.cols / 32) * 8) / 8;
        const unsigned long long *g = reinterpret_cast<const unsigned long long *>(E.img);
        for (int i = (int)threadIdx.x; i < words; i += TH)
            reinterpret_cast<unsigned long long *>(smem)[i] = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (trc) trc[3] = wall_clock64();
    if (gw >= P.total_rows) return;
    typename Fmt::X X;
    X.load(L.codes, L.scale, L.xsum, lane, P.nblk);
    float a[NM][RW];
#pragma unroll
    for (int i = 0; i < RW; i++)
#pragma unroll
        for (int m = 0; m < NM; m++) a[m][i] = w[m][i].dot(X);
#pragma unroll
    for (int i = 0; i < RW; i++)
#pragma unroll
        for (int m = 0; m < NM; m++) a[m][i] = wave_sum(a[m][i]);
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int i = 0; i < RW; i++) {
        if (lane == i) { a0 = a[0][i]; if constexpr (NM == 2) a1 = a[1][i]; }
    }
    const int v = lane * W + gw;
    if (lane < RW && v < P.total_rows) dec_finish_row<EPI>(P, dec_locate(P, v), a0, a1);
    if (trc) trc[4] = wall_clock64();
}

// host side (ifa_dwoffn_<format>.hip)
bool dec_wo_ffn_supported(int w_dtype, int wo_dtype, int w3_dtype, int dim, int wo_cols, int ffn_rows, bool glu, int num_cus);
int dec_wo_ffn_launch(int w_dtype, bool glu, const DecGemvParams &PW, const DecGemvParams &P, const DecWoFfnExtra &E, int num_cus, hipStream_t s);
template <int DT>
int dec_wo_ffn_launch_dt(bool glu, const DecGemvParams &PW, const DecGemvParams &P, const DecWoFfnExtra &E, int num_cus, hipStream_t s);

} // namespace ifa
