// ifa_gemm_rows_gemv.h -- 2..4 activation rows against a streamed 4-bit matrix: the batched decode step (and 2..4-token prompts)
// of a few queries, with the STRUCTURE of the batch-1 GEMV (ifa_decode_kernels.h) instead of the matrix-core tiles of
// ifa_gemm_rows_mfma*.h.  Round 4: 1 -> 2 queries cost +42 % per step (1.33 -> 1.89 ms) because the MFMA kernel splits K over
// the 8 waves of a workgroup, stages 8 rows whatever T is, and ends in a cross-wave reduction behind two barriers -- fixed phases
// of ~8 us per launch around 3-5 us of streaming.  Here a wave owns whole rows (no cross-wave sums, no barrier after the
// prologue), the first weight requests leave before the rows are staged, and every row of the matrix is read exactly once.
//
// Arithmetic = the reference's T > 1 branch (dequantised F16 weights x F16 activations, fp32 accumulation, F16 result:
// MatrixMultiplication -> DequantizeTensor + GemmEx, inference_worker.cc:2374-2415): w = half(fma(q, scale, base)) exactly as
// the MFMA kernels (ifa_dequant_q4.h), products of halves summed in fp32 by v_dot2_f32_f16 (lane-local over the lane's blocks,
// then the wave butterfly) -- another fp32 summation order than the MFMA tiles, the same tolerance against the oracle
// (tests/test_gpu_engine.py).  Weights: the tiled rows the batch-1 GEMV streams (row = nblk 16-byte code blocks, then nblk
// (base, scale) words) -- no third copy.  Formats: Q4_B32T1A / B.
//
// LDS image of the T rows: piece (block b, quarter s) of row t at ((b / 64 * 4 + s) * 64 + b % 64) * 16 + t * NJ * 4096: the 64
// lanes of a wave read 1024 contiguous bytes per request (lane l owns blocks l, l + 64, ...), conflict-free; a row's image is
// NJ * 4096 bytes (rows shorter than 64 NJ blocks leave the tail unused).
#pragma once
#include "ifa_decode_kernels.h"
#include "ifa_gemm_rows_mfma.h"
#include "ifa_dequant_q4.h"

namespace ifa {

constexpr int GV_THREADS = 512, GV_WAVES = 8;

// virtual row -> (set's weights, bias, output matrix and stride, row inside the set); v is wave-uniform
struct GvRow { const uint8_t *w; const uint8_t *w1; const half_t *b0; half_t *y; int ldy; int row; };
__device__ __forceinline__ GvRow gv_locate(const GmArgs &P, int v)
{
    const int r0 = P.rows[0], r1 = P.rows[1];
    const bool in1 = P.nsets > 1 && v >= r0, in2 = P.nsets > 2 && v >= r0 + r1;
    const bool per_set = P.Yset[0] != nullptr;
    GvRow g;
    const int set = in2 ? 2 : (in1 ? 1 : 0);
    g.row = v - (in2 ? r0 + r1 : (in1 ? r0 : 0));
    g.w = set == 2 ? P.W[2] : (set == 1 ? P.W[1] : P.W[0]);
    g.w1 = P.W1;
    g.b0 = set == 2 ? P.bias[2] : (set == 1 ? P.bias[1] : P.bias[0]);
    if (per_set) {
        g.y = set == 2 ? P.Yset[2] : (set == 1 ? P.Yset[1] : P.Yset[0]);
        g.ldy = set == 2 ? P.ldyset[2] : (set == 1 ? P.ldyset[1] : P.ldyset[0]);
    } else {
        g.y = P.Y + (v - g.row);
        g.ldy = P.ldy;
    }
    return g;
}

// NJ: blocks per lane (ceil(nblk / 64)); RW: rows (GLU: row pairs) of a wave in flight; T: activation rows (2..4)
template <int NJ, int RW, int EPI, int NORM, int T>
__global__ void __launch_bounds__(GV_THREADS, 2) k_rows_gemv(const GmArgs P)
{
    constexpr bool GLU = EPI == GM_GLU;
    constexpr int NM = GLU ? 2 : 1;                  // matrices per virtual row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = P.nblk, K = nblk * 32, chunks = nblk * 4;
    const size_t row_bytes = (size_t)nblk * 20;
    const int gw = blockIdx.x * GV_WAVES + wave, W = gridDim.x * GV_WAVES;
    const int nbatch = (P.total_rows + RW - 1) / RW;
    const float up = q4_fp8_up();
    constexpr size_t IMG = (size_t)NJ * 4096;          // bytes of one row's image

    WRowQ4<NJ> cur[RW][NM];
    auto load_batch = [&](WRowQ4<NJ> (&dst)[RW][NM], int b) {
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            const int v = min(min(b, nbatch - 1) * RW + rr, P.total_rows - 1);       // clamped, unconditional (see WRowQ4::load)
            const GvRow g = gv_locate(P, v);
            dst[rr][0].load(g.w + (size_t)g.row * row_bytes, nblk, lane);
            if constexpr (GLU) dst[rr][1].load(g.w1 + (size_t)g.row * row_bytes, nblk, lane);
        }
    };
    load_batch(cur, gw);                 // the first rows are on their way before the activation rows are staged

    // ---- stage the T rows (RMS-normalised when NORM): piece c (8 halves) of a row is thread c % 512's request number c / 512 --
    // all requests first, one pass: with the norm, piece c is lane c % 64 of group c / 64 of the canonical sum of squares
    // (ifa_math.h), i.e. request k of wave w belongs to group w + 8 k
    float *part = reinterpret_cast<float *>(smem + (size_t)T * IMG);               // [T][ngroups] group sums of squares
    const int ngroups = (chunks + 63) >> 6;
    auto dest = [&](int c, int t) { const int b = c >> 2, s = c & 3; return smem + (size_t)t * IMG + ((size_t)(((b >> 6) * 4 + s) * 64 + (b & 63))) * 16; };
    constexpr int XK = (NJ + 1) / 2;                                                // passes: chunks <= 256 NJ
    u32x4 xv[T][XK];
    u32x4 nwv[XK];
#pragma unroll
    for (int k = 0; k < XK; k++) {
        const int c = min(tid + GV_THREADS * k, chunks - 1);
#pragma unroll
        for (int t = 0; t < T; t++) xv[t][k] = *reinterpret_cast<const u32x4 *>(P.X + (size_t)t * P.ldx + (size_t)c * 8);
        if constexpr (NORM == 1) nwv[k] = *reinterpret_cast<const u32x4 *>((P.norm_w ? P.norm_w : P.X) + (size_t)c * 8);
    }
    if constexpr (NORM == 1) {
#pragma unroll
        for (int k = 0; k < XK; k++) {
            const bool in = tid + GV_THREADS * k < chunks;
            float pg[T];
#pragma unroll
            for (int t = 0; t < T; t++) {
                rms_h8 v8 = __builtin_bit_cast(rms_h8, xv[t][k]);
                if (!in) {
#pragma unroll
                    for (int i = 0; i < 8; i++) v8[i] = (half_t)0;
                }
                pg[t] = rms_chunk_sq(v8);
            }
#pragma unroll
            for (int t = 0; t < T; t++) pg[t] = wave_sum(pg[t]);
            if (lane == 0 && wave + GV_WAVES * k < ngroups) {
#pragma unroll
                for (int t = 0; t < T; t++) part[t * ngroups + wave + GV_WAVES * k] = pg[t];
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; t++) {
            const float scale = rms_scale_of(rms_total(part + t * ngroups, ngroups), K, P.eps);
#pragma unroll
            for (int k = 0; k < XK; k++) {
                rms_h8 xh = __builtin_bit_cast(rms_h8, xv[t][k]);
                const rms_h8 wv = __builtin_bit_cast(rms_h8, nwv[k]);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float v = (float)xh[i] * scale;
                    if (P.norm_w) { const float mlt = P.multi_base + (float)wv[i]; v = v * mlt; }
                    xh[i] = f2h(v);
                }
                xv[t][k] = __builtin_bit_cast(u32x4, xh);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < XK; k++) {
        const int c = tid + GV_THREADS * k;
        if (c < chunks) {
#pragma unroll
            for (int t = 0; t < T; t++) *reinterpret_cast<u32x4 *>(dest(c, t)) = xv[t][k];
        }
    }
    __syncthreads();

    // ---- the rows of this wave, RW at a time, the next batch requested before the current one is computed
    for (int b = gw; b < nbatch; b += W) {
        WRowQ4<NJ> nxt[RW][NM];
        if (b + W < nbatch) load_batch(nxt, b + W);       // (wave-uniform; most launches give a wave one batch)
        float acc[RW][NM][T];
#pragma unroll
        for (int rr = 0; rr < RW; rr++)
#pragma unroll
            for (int mm = 0; mm < NM; mm++)
#pragma unroll
                for (int t = 0; t < T; t++) acc[rr][mm][t] = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const bool ok = lane + 64 * j < nblk;                    // lanes past the row end: their clamped block contributes 0
#pragma unroll
            for (int s = 0; s < 4; s++) {
                u32x4 xs[T];
#pragma unroll
                for (int t = 0; t < T; t++) {
                    xs[t] = *reinterpret_cast<const u32x4 *>(smem + (size_t)t * IMG + (size_t)((j * 4 + s) * 64 + lane) * 16);
                    if (!ok) xs[t] = u32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int rr = 0; rr < RW; rr++)
#pragma unroll
                    for (int mm = 0; mm < NM; mm++) {
                        const uint32_t sbw = cur[rr][mm].sb[j];
                        const float base = hbits2f((uint16_t)(sbw & 0xFFFFu)), scale_up = hbits2f((uint16_t)(sbw >> 16)) * up;
                        q4_h2 w[4];
                        q4x8_dequant(cur[rr][mm].c[j][s], scale_up, base, w);
#pragma unroll
                        for (int t = 0; t < T; t++)
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                acc[rr][mm][t] = __builtin_amdgcn_fdot2(w[i], __builtin_bit_cast(q4_h2, xs[t][i]), acc[rr][mm][t], false);
                    }
            }
        }
        // ---- wave sums; lane t finishes token t of every row of the batch
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            const int v = b * RW + rr;
            float mine[NM];
#pragma unroll
            for (int mm = 0; mm < NM; mm++) {
                mine[mm] = 0.0f;
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const float sum = wave_sum(acc[rr][mm][t]);
                    if (lane == t) mine[mm] = sum;
                }
            }
            if (v < P.total_rows && lane < T) {
                const GvRow g = gv_locate(P, v);
                half_t y = f2h(mine[0]);
                if (g.b0) y = f2h(h2f(y) + h2f(g.b0[g.row]));
                if constexpr (EPI == GM_RESIDUAL) {
                    y = f2h(h2f(P.res[(size_t)lane * P.ldres + v]) + h2f(y));                // TensorOpr::Add (half add)
                } else if constexpr (GLU) {
                    half_t y3 = f2h(mine[NM - 1]);
                    if (P.bias1) y3 = f2h(h2f(y3) + h2f(P.bias1[g.row]));
                    const half_t act = f2h(act_fn(h2f(y), P.act_kind));                      // TensorOpr::Activation -> F16
                    y = f2h(h2f(act) * h2f(y3));                                             // TensorOpr::Mul
                }
                g.y[(size_t)lane * g.ldy + g.row] = y;
            }
        }
#pragma unroll
        for (int rr = 0; rr < RW; rr++)
#pragma unroll
            for (int mm = 0; mm < NM; mm++) cur[rr][mm] = nxt[rr][mm];
    }
}

// 2 <= T <= 4, Q4_B32T1A / B tiled rows (P.mo == 0), cols % 32 == 0 and <= 16384; same arguments as gemm_rows_mfma_launch
bool gemm_rows_gemv_ok(const GmArgs &P, int epi, int norm);
int gemm_rows_gemv_launch(const GmArgs &P, int epi, int norm, hipStream_t s);

} // namespace ifa
