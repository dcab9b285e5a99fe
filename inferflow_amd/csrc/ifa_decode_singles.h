// ifa_decode_singles.h -- the single-row experts of a batched mixture-of-experts step (round 4).
// A step of 8 queries x top-2 spreads 16 entries over 8 experts: two or three of them get exactly one row.  The reference
// runs such a row through its T = 1 branch (TensorOpr::Quantize + Gemv_AX, inference_worker.cc:1772-1774: Q8 activations,
// int8 dot, the arithmetic of the decode GEMV); round 3 ran them as three grouped launches over the reference-layout blocks
// (k_gemv_ax8_grouped: 3.9 TB/s) behind two quantiser launches and an element-wise one.  Here they take the batch-1 decode
// kernels' structure on the tiled rows those kernels stream: blockIdx.y is a slot of the device-built list (k_moe_build's
// MoeSingle {expert, entry}); absent slots leave at once; the workgroup quantises its entry's row (XPre<0>: the same Q8_B32T2
// quantiser as the decode step), streams the expert's rows and, for w1 / w3, writes act(w1 x) * (w3 x) -- two launches
// (gated pair, w2) instead of six.  Same per-row expression as ax8_term / WRowQ4::dot.  Q4_B32T1A / B.
#pragma once
#include "ifa_decode_kernels.h"
#include "ifa_moe.h"

namespace ifa {

struct DecSinglesParams {
    const MoeSingle *singles;
    const int *counts;                 // counts[2] = singles in the list
    const uint8_t *const *wtab;        // [expert][4] tiled pointers {w1, w3, w2, -}
    int which;                         // table column of the matrix (GLU: which, which + 1)
    const half_t *X; int ldx;          // gathered activations [entries][cols]
    half_t *Y; int ldy;                // outputs [entries][rows]
    int rows, cols, nblk, act_kind;
};

template <int DT, int NJ, int RW, bool GLU>
__global__ void __launch_bounds__(DEC_THREADS) k_dec_singles(const DecSinglesParams S)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.y >= S.counts[2]) return;                                    // (uniform over the workgroup)
    const int e = __builtin_amdgcn_readfirstlane(S.singles[blockIdx.y].expert), pos = __builtin_amdgcn_readfirstlane(S.singles[blockIdx.y].pos);
    const half_t *x = S.X + (size_t)pos * S.ldx;
    half_t *y = S.Y + (size_t)pos * S.ldy;
    constexpr int MAXC = (NJ * 8 * block_capacity(DT) + DEC_THREADS - 1) / DEC_THREADS;
    XPre<0, MAXC> pre;
    pre.issue(x, nullptr, nullptr, S.cols);                                       // the row first: the CU's queue is FIFO across waves
    const uint8_t *w0 = S.wtab[4 * e + S.which], *w1 = GLU ? S.wtab[4 * e + S.which + 1] : nullptr;
    const XLds L = xlds_carve(smem, S.cols);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * DEC_WAVES + wave, W = gridDim.x * DEC_WAVES;
    using Fmt = DecFmt<DT, NJ>;
    const size_t row_bytes = tiled_row_bytes(DT, (size_t)S.nblk);
    constexpr int NM = GLU ? 2 : 1;
    const int npass = (S.rows + RW * W - 1) / (RW * W);
    typename Fmt::W w[NM][RW];
    auto load_pass = [&](int pass) {
#pragma unroll
        for (int i = 0; i < RW; i++) {
            const int v = (pass * RW + i) * W + gw;
            if (i > 0 && v >= S.rows) continue;                                    // (rows past the end are not requested; row 0 of a pass clamped)
            const int row = min(v, S.rows - 1);
            w[0][i].load(w0 + (size_t)row * row_bytes, S.nblk, lane);
            if constexpr (GLU) w[1][i].load(w1 + (size_t)row * row_bytes, S.nblk, lane);
        }
    };
    __syncthreads();                                                              // every wave's activation request is queued
    load_pass(0);
    pre.finish(nullptr, nullptr, 0.0f, 0.0f, S.cols, L, nullptr);
    if (gw >= S.rows) return;
    typename Fmt::X X;
    X.load(L.codes, L.scale, L.xsum, lane, S.nblk);
    for (int pass = 0; pass < npass; pass++) {
        if (pass > 0) load_pass(pass);
        float a[NM][RW];
#pragma unroll
        for (int i = 0; i < RW; i++)
#pragma unroll
            for (int mm = 0; mm < NM; mm++) a[mm][i] = w[mm][i].dot(X);
#pragma unroll
        for (int i = 0; i < RW; i++)
#pragma unroll
            for (int mm = 0; mm < NM; mm++) a[mm][i] = wave_sum(a[mm][i]);
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int i = 0; i < RW; i++)
            if (lane == i) { a0 = a[0][i]; if constexpr (GLU) a1 = a[1][i]; }
        const int v = (pass * RW + lane) * W + gw;
        if (lane < RW && v < S.rows) {
            half_t out = f2h(a0);
            if constexpr (GLU) {
                const half_t t2 = f2h(a1);
                const half_t act = f2h(act_fn(h2f(out), S.act_kind));           // TensorOpr::Activation -> F16
                out = f2h(h2f(act) * h2f(t2));                                    // TensorOpr::Mul
            }
            y[v] = out;
        }
    }
}

// Q4_B32T1A / B tiled expert tables; cols % 32 == 0 and <= 16384 (GLU: <= 8192)
bool dec_singles_supported(int w_dtype, size_t rows, size_t cols, bool glu);
int dec_singles_launch(int w_dtype, const DecSinglesParams &S, bool glu, int max_singles, hipStream_t s);

} // namespace ifa
