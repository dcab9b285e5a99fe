// ifa_decode_wo_ffn.h -- round 4: the Wo rows IN FRONT of the W1 / W3 launch (one launch instead of two).
//
// Reference ops (batch 1): Quantize(attention output) -> Wo GEMV -> +bias -> Add(residual)           (inference_worker.cc:1339-1404)
//                          -> RmsNorm -> Quantize -> W1 GEMV, W3 GEMV -> activation -> Mul             (:1660-1923)
// with the arithmetic of k_dec_gemv<EPI_RESIDUAL, NORM 2> and k_dec_gemv<EPI_GLU, NORM 1> (shared row / prologue / epilogue code:
// fused == separate launches bit for bit, tests/test_gpu_fused_chain.py).
//
// Why this order of fusion pays where "Wo behind the attention" did not: a hand-off between workgroups
// costs two or three memory round trips whichever way it is built -- what matters is whether anything useful happens during
// them.  Here the SHORT op (Wo: 10 MB, a 2.5 us chain of request / first byte / dot) sits in front of the LONG stream (W1 | W3:
// 56 MB, 8.5 us at the rate the chip delivers): while half of the waves compute the Wo rows, publish them as {tag, half}
// granules, gather all 4096 of them from the other workgroups, normalise and quantise, the other half already have ALL their
// W1 / W3 rows in flight; the gather and the quantiser end (~5 us) long before the weight stream does (~11 us), so the whole Wo
// launch -- boundary, 1 us to its first request, 1.2 us to its first byte, its 1.6 us of streaming alone -- disappears under it.
//
// Waves [0, NP): "front" waves -- Wo rows (from the quantised attention output the previous launch left, no prologue), the
// all-gather of a = x + Wo.att, the FFN norm + quantiser (XPre<.., NT>: LDS counters, no s_barrier), then their own W1 / W3 rows.
// Waves [NP, 16): request all their W1 / W3 rows at once and wait for the LDS image.
// Hand-off (form R1 of the guide's price list): the Wo rows are stored WRITE-THROUGH (agent-scope stores) into the ordinary
// activation buffer, each front wave drains its stores (vmcnt(0)), the workgroup's last one posts ONE flag granule {tag, 1};
// wave 0 of every workgroup polls the 256 flags (2 KB per round -- polling the 32 KB of per-element granules instead was
// 8 MB per round over the grid and starved the weight stream: 7.4 us for the gather), then every front thread reads its 16
// bytes of `a` past the caches.  tag = (decode call, position) as in ifa_decode_qkv_attn.h; bounded waits, error word.
#pragma once
#include "ifa_decode_kernels.h"

namespace ifa {

struct DecWoFfnExtra {
    unsigned long long *a_flags;    // this layer's done flags, one {tag, 1} granule per workgroup: its Wo rows are in memory
    const int *state;               // state[1] = position of the step (tag)
    const unsigned *epoch;          // device word: decode-call counter (tag)
    unsigned epoch_add;
    unsigned *err;
    int timeout_us;
    long long *trace;               // optional [grid][8] wall-clock stamps (100 MHz): see tools/trace_fused.py
};

constexpr int WF_THREADS = 1024, WF_NP = 4;       // 4 front waves, 12 loader waves (round 4 trace: with 8 + 8 the front waves' half of
                                                  // the W1 / W3 rows was requested at 12 us, long after the loaders' half had drained)

// RW: row pairs of a LOADER wave (rows [0, RW * loader waves) of the matrix, strided); RWF: of a front wave (the rest); RWO: Wo rows
// of a front wave
template <int DT, int NJ, int RW, int RWF, int RWO, int EPI>
__global__ void __launch_bounds__(WF_THREADS) k_dec_wo_ffn(const half_t *pxq, const half_t *pnw, const half_t *pnb, int pcols,
                                                           const DecGemvParams PW, const DecGemvParams P, const DecWoFfnExtra E)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long t_kernel = wall_clock64();
    long long *const trf = (E.trace && threadIdx.x == 0) ? E.trace + (size_t)blockIdx.x * 8 : nullptr;          // a front wave
    long long *const trl = (E.trace && threadIdx.x == WF_NP * 64) ? E.trace + (size_t)blockIdx.x * 8 : nullptr;  // a loader wave
    if (trf) trf[0] = t_kernel;
    static_assert(EPI == EPI_GLU || EPI == EPI_ACT, "FFN up-projection epilogues");
    constexpr int TH = WF_THREADS, NP = WF_NP, PT = NP * 64;
    constexpr int NM = EPI == EPI_GLU ? 2 : 1;
    constexpr int MAXC = (NJ * 8 * block_capacity(DT) + PT - 1) / PT;
    using Fmt = DecFmt<DT, NJ>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool front = wave < NP;
    const int chunks = pcols >> 3;
    XPre<1, MAXC, false, PT> pre;
    typename Fmt::X XW;                    // front waves: the quantised attention output (Wo's activation)
    typename Fmt::W wo[RWO];
    half_t wres = (half_t)0;
    const int gwo = (int)blockIdx.x * NP + wave, WWO = (int)gridDim.x * NP;         // Wo rows over the front waves of the grid
    const size_t wo_row_bytes = tiled_row_bytes(DT, (size_t)PW.nblk);
    if (front) {
        // requests of the front waves, in the order they are needed: the image of the attention output (written by the previous
        // launch), the Wo rows, their residual values, the FFN norm weights of this thread's chunk(s)
        const XqImage Q = xq_image_carve(const_cast<half_t *>(pxq), PW.cols);
        XW.load(Q.codes, Q.scale, Q.xsum, lane, PW.nblk);
#pragma unroll
        for (int i = 0; i < RWO; i++) {
            const int v = i * WWO + gwo;
            if (i > 0 && v >= PW.total_rows) continue;
            wo[i].load(PW.W0[0] + (size_t)min(v, PW.total_rows - 1) * wo_row_bytes, PW.nblk, lane);
        }
        wres = PW.residual[min(min(lane, RWO - 1) * WWO + gwo, PW.total_rows - 1)];
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            const int c = (int)threadIdx.x + k * PT;
            if (c < chunks) {
                if (pnw) pre.wv[k] = *reinterpret_cast<const half8_t *>(pnw + (size_t)c * 8);
                if (pnb) pre.bv[k] = *reinterpret_cast<const half8_t *>(pnb + (size_t)c * 8);
            }
        }
    }
    const int pos = *(const __attribute__((address_space(4))) int *)(E.state + 1);
    const unsigned epoch = ((*(const __attribute__((address_space(4))) unsigned *)(E.epoch) + E.epoch_add) << 20) | ((unsigned)pos & 0xFFFFFu);
    const XLds L = xlds_carve(smem, pcols);
    // ---- the W1 (| W3) rows: the loader waves deal rows [0, RW * WL) among themselves (strided over the grid's loader waves),
    // the front waves the rest -- they request theirs last, so they get few
    constexpr int NL = TH / 64 - NP;
    const int WL = (int)gridDim.x * NL, WF = (int)gridDim.x * NP;
    const int W = front ? WF : WL;                                                            // stride of this wave's rows
    const int gw = front ? (int)blockIdx.x * NP + wave : (int)blockIdx.x * NL + (wave - NP);  // its index among its kind
    const int row0 = front ? min(RW * WL, P.total_rows) : 0;                                  // first row of its kind
    const int row1 = front ? P.total_rows : min(RW * WL, P.total_rows);                       // end of its kind's rows
    const size_t row_bytes = tiled_row_bytes(DT, (size_t)P.nblk);
    constexpr int RMAX = RW > RWF ? RW : RWF;
    typename Fmt::W w[NM][RMAX];
    auto load_rows = [&](int i0, int i1) {
        const bool full = row0 + (RMAX - 1) * W + gw < row1;
        auto one = [&](int i) {
            const int v = min(row0 + i * W + gw, P.total_rows - 1);
            w[0][i].load(P.W0[0] + (size_t)v * row_bytes, P.nblk, lane);
            if constexpr (NM == 2) w[1][i].load(P.W1 + (size_t)v * row_bytes, P.nblk, lane);
        };
        if (full) {
#pragma unroll
            for (int i = 0; i < RMAX; i++) { if (i < i0 || i >= i1) continue; one(i); }
        } else {
#pragma unroll
            for (int i = 0; i < RMAX; i++) {
                if (i < i0 || i >= i1) continue;
                if (i > 0 && row0 + i * W + gw >= row1) continue;
                one(i);
            }
        }
    };
    if (threadIdx.x == TH - 1) { L.part[128] = 0.0f; L.part[129] = 0.0f; L.part[130] = 0.0f; L.part[131] = 0.0f; }
    if (!front) load_rows(0, 1);
    __syncthreads();
    if (!front) {
        // PACED: one row pair per wave in flight (8 waves x 5 KB = 40 KB per CU: what the memory pipeline takes without a backlog,
        // and enough for its full rate) until the front waves have their image.  All rows at once -- k_dec_gemv's way -- parks
        // ~100 KB of requests in the CU's issue queue, and the front waves' granule polls wait behind them: 4 us per poll round
        // trip, the fused launch took 27.6 us against 16.9 for the two launches (r04, gpurun_out/r4r).
#pragma unroll
        for (int i = 1; i < RW; i++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            load_rows(i, i + 1);
        }
        if (trl) trl[5] = wall_clock64();
    } else {
        // ---- Wo rows: dot, + bias, + residual (TensorOpr::Add, half), stored for the W2 launch's residual AND published
        float aw[RWO];
#pragma unroll
        for (int i = 0; i < RWO; i++) aw[i] = wo[i].dot(XW);
#pragma unroll
        for (int i = 0; i < RWO; i++) aw[i] = wave_sum(aw[i]);
        float a0 = 0.0f;
#pragma unroll
        for (int i = 0; i < RWO; i++) { if (lane == i) a0 = aw[i]; }
        const int v = lane * WWO + gwo;
        if (lane < RWO && v < PW.total_rows) {
            half_t y = dec_bias(a0, PW.b0[0], v);
            y = f2h(h2f(wres) + h2f(y));
            __hip_atomic_store(reinterpret_cast<uint16_t *>(PW.y[0]) + v, __builtin_bit_cast(uint16_t, y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's rows are in memory
        if (lane == 0) lds_counter_add(L.part + 129);
        if (trf) trf[1] = wall_clock64();
        const long long t_give_up = wall_clock64() + (long long)E.timeout_us * 100;
        if (wave == 0) {
            lds_counter_wait(L.part + 129, NP);
            if (lane == 0) __hip_atomic_store(E.a_flags + blockIdx.x, ((unsigned long long)epoch << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // every workgroup's flag (the grid is one workgroup per CU: all of them are running)
            for (;;) {
                bool ok = true;
                for (int f = lane; f < (int)gridDim.x; f += 64)
                    ok &= (unsigned)(__hip_atomic_load(E.a_flags + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == epoch;
                if (__all(ok)) break;
                if (wall_clock64() > t_give_up) { if (lane == 0) atomicExch(E.err, 0x61u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) lds_counter_add(L.part + 128);
        }
        lds_counter_wait(L.part + 128, 1);
        // ---- this thread's chunk(s) of a, past the caches (an older copy of the buffer may sit in this XCD's L2)
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
        if (trf) trf[2] = wall_clock64();
        pre.finish(P.norm_w, P.norm_b, P.multi_base, P.eps, P.cols, L, P.xn_out, nullptr);
        if (trf) trf[3] = wall_clock64();
        load_rows(0, RWF);
        if (trf) trf[4] = wall_clock64();
    }
    lds_counter_wait(L.part + 131, NP);
    typename Fmt::X X;
    X.load(L.codes, L.scale, L.xsum, lane, P.nblk);
    const int nrows = front ? RWF : RW;          // (wave-uniform)
    float a[NM][RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; i++)
#pragma unroll
        for (int m = 0; m < NM; m++) a[m][i] = (i < nrows && (i == 0 || row0 + i * W + gw < row1)) ? w[m][i].dot(X) : 0.0f;
#pragma unroll
    for (int i = 0; i < RMAX; i++)
#pragma unroll
        for (int m = 0; m < NM; m++) a[m][i] = wave_sum(a[m][i]);
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int i = 0; i < RMAX; i++) {
        if (lane == i) { a0 = a[0][i]; if constexpr (NM == 2) a1 = a[1][i]; }
    }
    const int v = row0 + lane * W + gw;
    if (lane < nrows && v < row1) dec_finish_row<EPI>(P, dec_locate(P, v), a0, a1);
    if (trf) trf[6] = wall_clock64();
    if (trl) trl[7] = wall_clock64();
}

// host side (ifa_dwoffn_<format>.hip)
bool dec_wo_ffn_supported(int w_dtype, int wo_dtype, int w3_dtype, int dim, int wo_cols, int ffn_rows, bool glu, int num_cus);
int dec_wo_ffn_launch(int w_dtype, bool glu, const DecGemvParams &PW, const DecGemvParams &P, const DecWoFfnExtra &E, int num_cus, hipStream_t s);
template <int DT>
int dec_wo_ffn_launch_dt(bool glu, const DecGemvParams &PW, const DecGemvParams &P, const DecWoFfnExtra &E, int num_cus, hipStream_t s);

} // namespace ifa
