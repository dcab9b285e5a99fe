// ifa_gemm_rows.hip -- Y[T][N] = X[T][K] . W[N][K]^T for a HANDFUL of rows (2 <= T <= 8): dynamic batching of decode
// steps and very short prompts.
//
// Same contract as ifa_gemm (the reference's T > 1 branch: weights dequantised to half, half activations, fp32
// accumulation, one F16 rounding, bias as a half add: MatrixMultiplication, inference_worker.cc:2374-2415), but
// organised like the decode GEMV instead of MFMA tiles: at this size the layer is a pure weight stream, and a 32x32
// MFMA tile with 2-8 live rows cannot keep enough bytes in flight (measured 16 us per 4096^2 Q4 matrix against ~4 us
// of streaming).  One workgroup per CU, every wave owns RW rows per pass from the row-local plane layout
// (ifa_tiled.h), all T activation rows sit in LDS in a chunk-major image (lane-consecutive 16-byte reads), each
// weight block is dequantised once and multiplied into the T accumulators with v_fma_mix (half operands, fp32 sum).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "ifa_host.h"
#include "ifa_decode_kernels.h"
#include "ifa_moe.h"

namespace ifa {

typedef _Float16 half8r __attribute__((ext_vector_type(8)));
typedef _Float16 half2r __attribute__((ext_vector_type(2)));

constexpr int GR_THREADS = 512, GR_WAVES = 8;

// LDS image: 16-byte chunk q (0..3) of block b of token t at ((t * 4 + q) * nblk + b) * 16
// A wave walks its rows block-column by block-column (64 blocks = 2048 weights of each of its RW rows per step); the
// next step's 20 bytes per lane and row are requested before the current ones are multiplied.
template <int TB, int RW>
__device__ __forceinline__ void gemm_rows_chunk(const uint8_t *__restrict__ Wt, int rows, int nblk, const half_t *__restrict__ X, int T, int t0,
                                                const half_t *__restrict__ bias, half_t *__restrict__ Y, char *smem)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = blockIdx.x * GR_WAVES + wave, W = gridDim.x * GR_WAVES;
    const int K = nblk * 32;
    const size_t row_bytes = tiled_row_bytes(Q4_B32T1A, (size_t)nblk);
    // ---- activations first (the CU's memory queue is FIFO: see k_dec_gemv), then the first weights
    const int nchunks = TB * nblk * 4;
    for (int c = tid; c < nchunks; c += GR_THREADS) {
        const int t = c / (nblk * 4), rem = c - t * (nblk * 4);      // rem = b * 4 + q in the source row
        const int b = rem >> 2, q = rem & 3;
        const int tt = min(t0 + t, T - 1);                            // rows past T: duplicates, never stored
        const u32x4 v = *reinterpret_cast<const u32x4 *>(X + (size_t)tt * K + (size_t)rem * 8);
        *reinterpret_cast<u32x4 *>(smem + ((size_t)(t * 4 + q) * nblk + b) * 16) = v;
    }
    const int nj = (nblk + 63) / 64;
    const int npass = (rows + RW * W - 1) / (RW * W);
    const int nsteps = npass * nj;
    struct Slice { u32x4 c[RW]; uint32_t sb[RW]; };
    auto fetch = [&](Slice &sl, int step) {
        const int pass = step / nj, j = step - pass * nj;
        const int blk = min(lane + 64 * j, nblk - 1);
#pragma unroll
        for (int i = 0; i < RW; i++) {
            const int v = min((pass * RW + i) * W + gw, rows - 1);
            const uint8_t *wrow = Wt + (size_t)v * row_bytes;
            sl.c[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow + (size_t)blk * 16));
            sl.sb[i] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(wrow + (size_t)nblk * 16 + (size_t)blk * 4));
        }
    };
    Slice cur, nxt;
    fetch(cur, 0);
    if (nsteps > 1) fetch(nxt, 1);
    __syncthreads();
    if (gw >= rows) return;              // (from this chunk: a grouped caller's next chunk starts with a barrier all waves reach)
    float acc[RW][TB];
#pragma unroll
    for (int i = 0; i < RW; i++)
#pragma unroll
        for (int t = 0; t < TB; t++) acc[i][t] = 0.0f;
    for (int step = 0; step < nsteps; step++) {
        const int pass = step / nj, j = step - pass * nj;
        const int blk = lane + 64 * j;
        const bool ok = blk < nblk;
        const int bc = min(blk, nblk - 1);
        float base[RW], scale[RW];
#pragma unroll
        for (int i = 0; i < RW; i++) {
            base[i] = hbits2f((uint16_t)(cur.sb[i] & 0xFFFFu));
            scale[i] = ok ? hbits2f((uint16_t)(cur.sb[i] >> 16)) : 0.0f;
            if (!ok) base[i] = 0.0f;                                  // lanes past the row end contribute exactly 0
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            half8r xv[TB];
#pragma unroll
            for (int t = 0; t < TB; t++) xv[t] = *reinterpret_cast<const half8r *>(smem + ((size_t)(t * 4 + q) * nblk + bc) * 16);
#pragma unroll
            for (int i = 0; i < RW; i++) {
                const uint32_t cw = cur.c[i][q];
                // byte b of the word: low nibble = element 8q+2b, high nibble = the next one.  Nibbles spread into bytes so
                // each value is one v_cvt_f32_ubyteN; the two halves of a byte and the matching activation pair go through
                // ONE v_dot2_f32_f16 (two half products + fp32 accumulate): half the multiply-adds of a v_fma_mix per value
                const uint32_t lo = cw & 0x0F0F0F0Fu, hi = (cw >> 4) & 0x0F0F0F0Fu;
                const float ql[4] = {ubyte_f32<0>(lo), ubyte_f32<1>(lo), ubyte_f32<2>(lo), ubyte_f32<3>(lo)};
                const float qh[4] = {ubyte_f32<0>(hi), ubyte_f32<1>(hi), ubyte_f32<2>(hi), ubyte_f32<3>(hi)};
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    half2r w2;
                    w2[0] = f2h(__builtin_fmaf(ql[b], scale[i], base[i]));   // the reference's dequantised half
                    w2[1] = f2h(__builtin_fmaf(qh[b], scale[i], base[i]));
#pragma unroll
                    for (int t = 0; t < TB; t++) {
                        half2r x2; x2[0] = xv[t][2 * b]; x2[1] = xv[t][2 * b + 1];
                        acc[i][t] = __builtin_amdgcn_fdot2(w2, x2, acc[i][t], false);
                    }
                }
            }
        }
        cur = nxt;
        if (step + 2 < nsteps) fetch(nxt, step + 2);
        if (j + 1 < nj) continue;
        // ---- end of a pass: reduce, lane (i * TB + t) finishes element (row i, token t)
        float mine = 0.0f;
#pragma unroll
        for (int i = 0; i < RW; i++)
#pragma unroll
            for (int t = 0; t < TB; t++) {
                const float r = wave_sum(acc[i][t]);
                if (lane == i * TB + t) mine = r;
                acc[i][t] = 0.0f;
            }
        if (lane < RW * TB) {
            const int i = lane / TB, t = lane % TB;
            const int row = (pass * RW + i) * W + gw;
            if (row < rows && t0 + t < T) {
                half_t y = f2h(mine);
                if (bias) y = f2h(h2f(y) + h2f(bias[row]));
                Y[(size_t)(t0 + t) * rows + row] = y;
            }
        }
    }
}

template <int TB, int RW>
__global__ void __launch_bounds__(GR_THREADS) k_gemm_rows_q4(const uint8_t *__restrict__ Wt, int rows, int nblk,
                                                             const half_t *__restrict__ X, int T, int t0,
                                                             const half_t *__restrict__ bias, half_t *__restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_rows_chunk<TB, RW>(Wt, rows, nblk, X, T, t0, bias, Y, smem);
}

// Mixture of experts, a handful of rows per expert (ifa_moe.h "smalls"): blockIdx.y is one expert's group of 2..8
// consecutive rows of the gathered activations; its tiled weights come from the pointer table.  Groups of more rows than
// the LDS image holds at this K (tb_cap) are walked in chunks (the weights are streamed once per chunk).
template <int RW>
__global__ void __launch_bounds__(GR_THREADS) k_gemm_rows_q4_grouped(const MoeSmallGroup grp, int rows, int nblk, const half_t *__restrict__ X,
                                                                     half_t *__restrict__ Y, int tb_cap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.y >= grp.counts[3]) return;
    const MoeTile g = grp.smalls[blockIdx.y];
    const uint8_t *Wt = grp.wtab_tiled[4 * g.expert + grp.which_tiled];
    const half_t *Xg = X + (size_t)g.row0 * nblk * 32;
    half_t *Yg = Y + (size_t)g.row0 * rows;
    for (int t0 = 0; t0 < g.nrows; t0 += tb_cap) {
        const int n = min(tb_cap, g.nrows - t0);
        if (t0 > 0) __syncthreads();             // the previous chunk's LDS image has been read
        if (n <= 2) gemm_rows_chunk<2, RW>(Wt, rows, nblk, Xg, g.nrows, t0, nullptr, Yg, smem);
        else if (n <= 4) gemm_rows_chunk<4, RW>(Wt, rows, nblk, Xg, g.nrows, t0, nullptr, Yg, smem);
        else gemm_rows_chunk<8, RW>(Wt, rows, nblk, Xg, g.nrows, t0, nullptr, Yg, smem);
    }
}

} // namespace ifa

using namespace ifa;

static int gr_num_cus()
{
    static int n = 0;
    if (!n) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

namespace ifa {      // ifa_gemm_rows_mfma.hip
bool gemm_rows_mfma_ok(size_t rows, size_t cols, size_t tokens);
int gemm_rows_mfma(const void *Wt_tiled, size_t rows, size_t cols, const void *x_f16, size_t tokens, const void *bias_f16, void *y_f16, hipStream_t s);
bool gemm_rows_use_mfma()
{
    static const bool fdot = getenv("IFA_ROWS_KERNEL") && !strcmp(getenv("IFA_ROWS_KERNEL"), "fdot");     // A/B switch for tools/
    return !fdot;
}
}

// Q4_B32T1 tiled weights only; returns IFA_ERR_STATE (untouched outputs) when the shape is not covered so that the
// caller can fall back to ifa_gemm.  2..16 rows on the matrix cores (ifa_gemm_rows_mfma.hip) when the row length is a
// multiple of 128, else 2..8 rows through the fdot2 kernel above.
extern "C" int ifa_gemm_rows_q4(const void *Wt_tiled, size_t rows, size_t cols, const void *x_f16, size_t tokens,
                                const void *bias_f16, void *y_f16, ifa_stream stream)
{
    IFA_REQUIRE(Wt_tiled && x_f16 && y_f16, "ifa_gemm_rows_q4: null pointer");
    // (2 rows: the fdot2 kernel is ~10 % faster -- 13.3 vs 14.7 us on a 12288 x 4096 matrix; from 3 rows on the matrix cores win)
    if (gemm_rows_use_mfma() && tokens >= 3 && gemm_rows_mfma_ok(rows, cols, tokens))
        return gemm_rows_mfma(Wt_tiled, rows, cols, x_f16, tokens, bias_f16, y_f16, ifa_s(stream));
    if (tokens < 2 || tokens > 8 || cols % 32 != 0 || rows == 0) return IFA_ERR_STATE;
    const int nblk = (int)(cols / 32);
    hipStream_t s = ifa_s(stream);
    int wgs = std::min(gr_num_cus(), (int)((rows + GR_WAVES - 1) / GR_WAVES));
    if (wgs < 1) wgs = 1;
    // tokens per launch: the LDS image is TB * cols * 2 bytes (<= 128 KiB)
    int tb = tokens <= 2 ? 2 : (tokens <= 4 ? 4 : 8);
    while ((size_t)tb * cols * 2 > 128 * 1024 && tb > 2) tb /= 2;
    if ((size_t)tb * cols * 2 > 128 * 1024) return IFA_ERR_STATE;
    for (size_t t0 = 0; t0 < tokens; t0 += (size_t)tb) {
        const size_t smem = (size_t)tb * cols * 2;
#define IFA_GR(TBV, RWV) { auto kern = k_gemm_rows_q4<TBV, RWV>; \
        if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        kern<<<dim3((unsigned)wgs), dim3(GR_THREADS), smem, s>>>((const uint8_t *)Wt_tiled, (int)rows, nblk, (const half_t *)x_f16, (int)tokens, (int)t0, \
                                                                  (const half_t *)bias_f16, (half_t *)y_f16); }
        if (tb == 2) IFA_GR(2, 2) else if (tb == 4) IFA_GR(4, 2) else IFA_GR(8, 2)
#undef IFA_GR
        IFA_LAUNCH_CHECK();
    }
    return IFA_OK;
}

namespace ifa {
// rows / cols of ONE expert matrix; X / Y: the gathered activations / outputs of all entries.  Returns IFA_ERR_STATE when
// the shape is not covered (the caller then leaves those experts to the MFMA tiles: small_max = 0).
int gemm_rows_q4_grouped_cap(size_t cols)
{
    if (cols % 32 != 0) return 0;
    int tb = 8;
    while ((size_t)tb * cols * 2 > 128 * 1024 && tb > 2) tb /= 2;
    return (size_t)tb * cols * 2 > 128 * 1024 ? 0 : tb;
}

int gemm_rows_q4_grouped(const MoeSmallGroup &grp, size_t rows, size_t cols, const void *X, void *Y, int max_groups, hipStream_t s)
{
    const int tb = gemm_rows_q4_grouped_cap(cols);
    if (tb == 0 || rows == 0) return ifa_fail(IFA_ERR_STATE, "grouped rows GEMM: %zu columns", cols);
    if (max_groups <= 0) return IFA_OK;
    // the experts share the chip: about two workgroups per CU over all groups
    int wgs = std::max(16, 2 * gr_num_cus() / max_groups);
    wgs = std::min(wgs, (int)((rows + GR_WAVES - 1) / GR_WAVES));
    const size_t smem = (size_t)tb * cols * 2;
    auto kern = k_gemm_rows_q4_grouped<2>;
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)wgs, (unsigned)max_groups), dim3(GR_THREADS), smem, s>>>(grp, (int)rows, (int)(cols / 32), (const half_t *)X, (half_t *)Y, tb);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}
}
