// ifa_gemv.hip -- TensorMul::Gemv_AX counterparts (src/tensor/tensor_mul.cu:766-846).
//
//  * int8 x intN path (reference: Gemv_AX8_* kernels, src/kernels/gemv.h:1499-1709):
//    weights in any of the 7 eligible block formats, x pre-quantised to Q8_B32T2.
//      y[r] = sum_blk xs_blk * ( scale_blk * dot(qw,qx) + base_blk * sum(qx) )
//    integer dots are exact (v_dot4_i32_i8); the fp32 combination is done per
//    block, lanes own blocks l, l+64, ... and a DPP/bpermute tree sums the wave.
//    The reference's 32-lane CUDA butterfly orders the fp32 adds differently;
//    results agree to fp32 round-off and are compared after rounding to half
//    with the tolerance stated in tests/ (DESIGN.md "Tolerances").
//  * fp16-activation path (gemv.h:469-1497): weights dequantised to half in
//    registers, fp32 accumulate (the reference accumulates three variants in
//    half, appendix A4; fp32 is at least as accurate).
//
// Kernels here are memory-bound: 1 wave per row (or row group), lanes stride the
// K dimension so every wave-level load covers a contiguous span.
#include "ifa_host.h"
#include "ifa_codec.h"
#include "ifa_tiled.h"
#include "ifa_moe.h"

namespace ifa {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ half_t finish_row(float acc, const half_t *bias, int row)
{
    half_t y = f2h(acc);
    if (bias) y = f2h(h2f(y) + h2f(bias[row]));   // half add, like TensorOpr::Add / gemv.h:524
    return y;
}

// x block k of a Q8_B32T2 row (34-byte AoS blocks)
__device__ __forceinline__ void load_x_block(const uint8_t *x, int k, int *xq, float &xs, int &xsum)
{
    RawBlock<34> xb;
    xb.load(x + (size_t)k * 34);
    float b0;
    decode_block<Q8_B32T2>(xb, xq, xs, b0);
    xsum = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) xsum += xq[i];
}

// fp32 contribution of one weight block (shared by every AX8 kernel so that
// all of them produce bit-identical results)
template <int DT>
__device__ __forceinline__ float ax8_term(int dot, int xsum, float scale, float base, float xs)
{
    float t = (float)dot * scale;
    if constexpr (DT != Q8_B32T2) {
        float u = (float)xsum * base;
        t = t + u;
    }
    return xs * t;
}

// ---------------------------------------------------------------- generic AX8
// Reference byte layout (AoS), any eligible format, any cols % cap == 0.
template <int DT, bool TILED>
__global__ void __launch_bounds__(256) k_gemv_ax8_generic(const uint8_t *__restrict__ W, int rows, int nblk,
                                                          const uint8_t *__restrict__ xq8,
                                                          const half_t *__restrict__ bias, half_t *__restrict__ y)
{
    constexpr int CAP = block_capacity(DT), BB = block_bytes(DT);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint8_t *wrow = W + (size_t)row * (TILED ? tiled_row_bytes(DT, (size_t)nblk) : (size_t)nblk * BB);
    float acc = 0.0f;
    for (int blk = lane; blk < nblk; blk += 64) {
        RawBlock<BB> b;
        if constexpr (TILED && DT == Q3H_B64T1) {
            // nibble pairs (ifa_tiled.h): the codes are read directly below, b stays unused
        } else if constexpr (TILED) {
            using L = TiledLayout<DT>;
#pragma unroll
            for (int p = 0; p < L::NPLANES; p++) {
                const uint16_t *s = reinterpret_cast<const uint16_t *>(wrow + (size_t)L::plane_start(p) * nblk
                                                                       + (size_t)blk * L::plane_len(p));
#pragma unroll
                for (int i = 0; i < L::plane_len(p) / 2; i++) b.w[L::plane_src_off(p) / 2 + i] = s[i];
            }
        } else {
            b.load(wrow + (size_t)blk * BB);
        }
        int q[CAP]; float scale, base;
        if constexpr (TILED && DT == Q3H_B64T1) {
            const uint8_t *p0 = wrow + (size_t)blk * 32, *p1 = wrow + (size_t)32 * nblk + (size_t)blk * 4;
#pragma unroll
            for (int i = 0; i < 32; i++) { const uint32_t d = p0[i]; q[2 * i] = (int)(d & 0x0F); q[2 * i + 1] = (int)(d >> 4); }
            base = hbits2f((uint16_t)(p1[0] | (p1[1] << 8))); scale = hbits2f((uint16_t)(p1[2] | (p1[3] << 8)));
        } else decode_block<DT>(b, q, scale, base);
#pragma unroll
        for (int h = 0; h < CAP / 32; h++) {
            int xq[32]; float xs; int xsum;
            load_x_block(xq8, blk * (CAP / 32) + h, xq, xs, xsum);
            int dot = 0;
#pragma unroll
            for (int i = 0; i < 32; i++) dot += q[32 * h + i] * xq[i];
            acc = acc + ax8_term<DT>(dot, xsum, scale, base, xs);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) y[row] = finish_row(acc, bias, row);
}

// The same kernel over the single-row experts of a mixture-of-experts step (ifa_moe.h): blockIdx.y = index into the
// device-side list; the expert's weights come from the pointer table, x / y are the entry's row of the gathered
// (quantised) activation / output buffers.  Same loop, same ax8_term: bit-identical to the op-level GEMV.
template <int DT>
__global__ void __launch_bounds__(256) k_gemv_ax8_grouped(const MoeGroup grp, int rows, int nblk, const uint8_t *__restrict__ xq8_rows,
                                                          size_t xq_row_bytes, half_t *__restrict__ y_rows)
{
    constexpr int CAP = block_capacity(DT), BB = block_bytes(DT);
    if ((int)blockIdx.y >= grp.counts[2]) return;
    const MoeSingle sg = grp.singles[blockIdx.y];
    const uint8_t *W = grp.wtab[sg.expert * 3 + grp.which];
    const uint8_t *xq8 = xq8_rows + (size_t)sg.pos * xq_row_bytes;
    half_t *y = y_rows + (size_t)sg.pos * rows;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint8_t *wrow = W + (size_t)row * ((size_t)nblk * BB);
    float acc = 0.0f;
    for (int blk = lane; blk < nblk; blk += 64) {
        RawBlock<BB> b;
        b.load(wrow + (size_t)blk * BB);
        int q[CAP]; float scale, base;
        decode_block<DT>(b, q, scale, base);
#pragma unroll
        for (int h = 0; h < CAP / 32; h++) {
            int xq[32]; float xs; int xsum;
            load_x_block(xq8, blk * (CAP / 32) + h, xq, xs, xsum);
            int dot = 0;
#pragma unroll
            for (int i = 0; i < 32; i++) dot += q[32 * h + i] * xq[i];
            acc = acc + ax8_term<DT>(dot, xsum, scale, base, xs);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) y[row] = finish_row(acc, nullptr, row);
}

int gemv_ax8_grouped(int w_dtype, const MoeGroup &grp, size_t rows, size_t cols, const void *xq8_rows, void *y_rows, int max_singles,
                     hipStream_t s)
{
    if (!ax8_eligible(w_dtype)) return ifa_fail(IFA_ERR_DTYPE, "grouped int8 GEMV: dtype %d", w_dtype);
    if (max_singles <= 0) return IFA_OK;
    const int cap = block_capacity(w_dtype);
    const size_t xrb = cols / 32 * 34;
    const dim3 grid(ifa_cdiv(rows, 4), (unsigned)max_singles);
    switch (w_dtype) {
#define IFA_GG(DTV) case DTV: k_gemv_ax8_grouped<DTV><<<grid, dim3(256), 0, s>>>(grp, (int)rows, (int)(cols / cap), (const uint8_t *)xq8_rows, xrb, (half_t *)y_rows); break;
    IFA_GG(Q8_B32T2) IFA_GG(Q6_B64T1) IFA_GG(Q5_B64T1) IFA_GG(Q4_B32T1A) IFA_GG(Q4_B32T1B) IFA_GG(Q4_B64T1) IFA_GG(Q3H_B64T1)
#undef IFA_GG
    default: return ifa_fail(IFA_ERR_DTYPE, "grouped int8 GEMV: dtype %d", w_dtype);
    }
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// ----------------------------------------------------- fast Q4_B32T1 (tiled/AoS)
// Lane l owns blocks l+64j of every row; x lives in registers (pre-split into
// even/odd nibble order so a packed dword of 8 codes needs 2 ANDs, 1 shift and
// 2 v_dot4).  R rows are in flight per wave: R*NJ dwordx4 + R*NJ dword loads.
template <int NJ, int R, bool TILED>
__global__ void __launch_bounds__(256) k_gemv_q4b32(const uint8_t *__restrict__ W, int rows, int nblk,
                                                    const uint8_t *__restrict__ xq8,
                                                    const half_t *__restrict__ bias, half_t *__restrict__ y,
                                                    int rows_per_wave)
{
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int row0 = gw * rows_per_wave;
    if (row0 >= rows) return;
    const int row_end = min(row0 + rows_per_wave, rows);
    const size_t row_bytes = TILED ? tiled_row_bytes(Q4_B32T1A, (size_t)nblk) : (size_t)nblk * 20;

    int xe[NJ][4], xo[NJ][4];
    float xs[NJ], xsf[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int blk = lane + 64 * j;
        xs[j] = 0.0f; xsf[j] = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; w++) { xe[j][w] = 0; xo[j][w] = 0; }
        if (blk < nblk) {
            const uint8_t *xb = xq8 + (size_t)blk * 34;
            xs[j] = hbits2f(*reinterpret_cast<const uint16_t *>(xb));
            int xsum = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t x0 = ld_u32_unaligned2(xb + 2 + 8 * w), x1 = ld_u32_unaligned2(xb + 2 + 8 * w + 4);
                xsum = sdot4(0x01010101, (int)x0, xsum);
                xsum = sdot4(0x01010101, (int)x1, xsum);
                xe[j][w] = (int)__builtin_amdgcn_perm(x1, x0, 0x06040200u);
                xo[j][w] = (int)__builtin_amdgcn_perm(x1, x0, 0x07050301u);
            }
            xsf[j] = (float)xsum;
        }
    }

    for (int r = row0; r < row_end; r += R) {
        u32x4 c[R][NJ];
        uint32_t sb[R][NJ];
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const uint8_t *wrow = W + (size_t)(r + rr) * row_bytes;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int blk = lane + 64 * j;
                const bool ok = (blk < nblk) && (r + rr < row_end);
                c[rr][j] = u32x4{0, 0, 0, 0}; sb[rr][j] = 0;
                if (ok) {
                    if constexpr (TILED) {
                        c[rr][j] = *reinterpret_cast<const u32x4 *>(wrow + (size_t)blk * 16);
                        sb[rr][j] = *reinterpret_cast<const uint32_t *>(wrow + (size_t)nblk * 16 + (size_t)blk * 4);
                    } else {
                        c[rr][j] = *reinterpret_cast<const u32x4_a4 *>(wrow + (size_t)blk * 20 + 4);
                        sb[rr][j] = *reinterpret_cast<const uint32_t *>(wrow + (size_t)blk * 20);
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                int dot = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t cw = c[rr][j][w];
                    dot = sdot4((int)(cw & 0x0F0F0F0Fu), xe[j][w], dot);
                    dot = sdot4((int)((cw >> 4) & 0x0F0F0F0Fu), xo[j][w], dot);
                }
                const float base = hbits2f((uint16_t)(sb[rr][j] & 0xFFFFu));
                const float scale = hbits2f((uint16_t)(sb[rr][j] >> 16));
                float t = (float)dot * scale;
                float u = xsf[j] * base;
                t = t + u;
                acc = acc + xs[j] * t;
            }
            acc = wave_sum(acc);
            if (lane == 0 && r + rr < row_end) y[r + rr] = finish_row(acc, bias, r + rr);
        }
    }
}

// ------------------------------------------------------------- fp16-x generic
template <int DT>
__global__ void __launch_bounds__(256) k_gemv_f16x_quant(const uint8_t *__restrict__ W, int rows, int nblk,
                                                         const half_t *__restrict__ x,
                                                         const half_t *__restrict__ bias, half_t *__restrict__ y)
{
    constexpr int CAP = block_capacity(DT), BB = block_bytes(DT);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint8_t *wrow = W + (size_t)row * nblk * BB;
    float acc = 0.0f;
    for (int blk = lane; blk < nblk; blk += 64) {
        RawBlock<BB> b;
        b.load(wrow + (size_t)blk * BB);
        int q[CAP]; float scale, base;
        decode_block<DT>(b, q, scale, base);
        const half_t *xp = x + (size_t)blk * CAP;
#pragma unroll
        for (int i = 0; i < CAP; i++) {
            float wv = h2f(f2h(block_value<DT>(q[i], scale, base)));  // half-rounded like the reference's arr_a[]
            acc = __builtin_fmaf(wv, h2f(xp[i]), acc);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) y[row] = finish_row(acc, bias, row);
}

// F16 weights (lm_head and every tensor below tensor_quant_threshold):
// 16 B per lane per step, x chunk in registers when cols <= 64*8*NJ.
template <int NJ>
__global__ void __launch_bounds__(256) k_gemv_f16w(const half_t *__restrict__ W, int rows, int cols,
                                                   const half_t *__restrict__ x, const half_t *__restrict__ bias,
                                                   half_t *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int chunks = cols >> 3;   // 8 halfs per chunk
    const u32x4 *wrow = reinterpret_cast<const u32x4 *>(W + (size_t)row * cols);
    const u32x4 *xv = reinterpret_cast<const u32x4 *>(x);
    float acc = 0.0f;
    if constexpr (NJ > 0) {
        u32x4 wreg[NJ], xreg[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int c = lane + 64 * j;
            wreg[j] = u32x4{0, 0, 0, 0}; xreg[j] = u32x4{0, 0, 0, 0};
            if (c < chunks) { wreg[j] = wrow[c]; xreg[j] = xv[c]; }
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) acc = dot8_f16(wreg[j], xreg[j], acc);
    } else {
        for (int c = lane; c < chunks; c += 64) {
            u32x4 wv = wrow[c], xx = xv[c];
            acc = dot8_f16(wv, xx, acc);
        }
    }
    // ragged tail (cols % 8)
    for (int c = (chunks << 3) + lane; c < cols; c += 64)
        acc = __builtin_fmaf(h2f(W[(size_t)row * cols + c]), h2f(x[c]), acc);
    acc = wave_sum(acc);
    if (lane == 0) y[row] = finish_row(acc, bias, row);
}

} // namespace ifa

using namespace ifa;

template <bool TILED>
static int launch_ax8(int w_dtype, const void *W, size_t rows, size_t cols, const void *xq8, const void *bias,
                      void *y, ifa_stream stream)
{
    int cap = block_capacity(w_dtype);
    size_t nblk = cols / (size_t)cap;
    hipStream_t s = ifa_s(stream);
    const uint8_t *Wp = (const uint8_t *)W; const uint8_t *xp = (const uint8_t *)xq8;
    const half_t *bp = (const half_t *)bias; half_t *yp = (half_t *)y;
    // fast path: Q4_B32T1, rows of 16-byte aligned planes
    if ((w_dtype == Q4_B32T1A || w_dtype == Q4_B32T1B) && nblk <= 512) {
        int nj = (int)((nblk + 63) / 64);
        int rpw = rows >= 8192 ? 4 : 2;
        unsigned waves = ifa_cdiv(rows, (size_t)rpw), grid = ifa_cdiv(waves, 4);
#define IFA_Q4_CASE(NJV) case NJV: k_gemv_q4b32<NJV, 2, TILED><<<dim3(grid), dim3(256), 0, s>>>(\
                                                      Wp, (int)rows, (int)nblk, xp, bp, yp, rpw); break;
        switch (nj) {
            IFA_Q4_CASE(1) IFA_Q4_CASE(2) IFA_Q4_CASE(3) IFA_Q4_CASE(4) IFA_Q4_CASE(5) IFA_Q4_CASE(6) IFA_Q4_CASE(7) IFA_Q4_CASE(8)
        default: return ifa_fail(IFA_ERR_ARG, "ifa_gemv: unexpected nj %d", nj);
        }
#undef IFA_Q4_CASE
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    unsigned grid = ifa_cdiv(rows, 4);
#define IFA_AX8_CASE(T) case T: k_gemv_ax8_generic<T, TILED><<<dim3(grid), dim3(256), 0, s>>>(\
                                                   Wp, (int)rows, (int)nblk, xp, bp, yp); break;
    switch (w_dtype) {
        IFA_AX8_CASE(Q8_B32T2) IFA_AX8_CASE(Q6_B64T1) IFA_AX8_CASE(Q5_B64T1) IFA_AX8_CASE(Q4_B32T1A)
        IFA_AX8_CASE(Q4_B32T1B) IFA_AX8_CASE(Q4_B64T1) IFA_AX8_CASE(Q3H_B64T1)
    default: return ifa_fail(IFA_ERR_DTYPE, "ifa_gemv: weight dtype %d is not eligible for the int8 path", w_dtype);
    }
#undef IFA_AX8_CASE
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

extern "C" {

int ifa_gemv(int w_dtype, const void *W, size_t rows, size_t cols, int x_dtype, const void *x,
             const void *bias_f16, void *y_f16, ifa_stream stream)
{
    IFA_REQUIRE(W && x && y_f16, "ifa_gemv: null pointer");
    int cap = block_capacity(w_dtype);
    IFA_REQUIRE(cap > 0 && w_dtype != F32, "ifa_gemv: unsupported weight dtype %d", w_dtype);
    if (rows == 0) return IFA_OK;
    IFA_REQUIRE(cols > 0 && cols % (size_t)cap == 0, "ifa_gemv: cols %zu not a multiple of block capacity %d", cols, cap);
    IFA_REQUIRE(rows < (1u << 30) && cols < (1u << 30), "ifa_gemv: shape too large");
    hipStream_t s = ifa_s(stream);
    if (x_dtype == Q8_B32T2) {
        // GemvCheckN, src/tensor/tensor_mul.cu:1101,1123,1186: cols % 32 (B32) / % 64 (B64)
        IFA_REQUIRE(ax8_eligible(w_dtype), "ifa_gemv: weight dtype %d cannot take Q8 activations", w_dtype);
        return launch_ax8<false>(w_dtype, W, rows, cols, x, bias_f16, y_f16, stream);
    }
    IFA_REQUIRE(x_dtype == F16, "ifa_gemv: x dtype %d (expected F16 or Q8_B32T2)", x_dtype);
    unsigned grid = ifa_cdiv(rows, 4);
    const half_t *xp = (const half_t *)x; const half_t *bp = (const half_t *)bias_f16; half_t *yp = (half_t *)y_f16;
    if (w_dtype == F16) {
        size_t chunks = cols / 8;
        const half_t *Wp = (const half_t *)W;
        if (cols % 8 == 0 && chunks <= 64 * 8) {
            int nj = (int)((chunks + 63) / 64);
#define IFA_F16_CASE(NJV) case NJV: k_gemv_f16w<NJV><<<dim3(grid), dim3(256), 0, s>>>(Wp, (int)rows, (int)cols, xp, bp, yp); break;
            switch (nj) {
                IFA_F16_CASE(1) IFA_F16_CASE(2) IFA_F16_CASE(3) IFA_F16_CASE(4) IFA_F16_CASE(5) IFA_F16_CASE(6) IFA_F16_CASE(7) IFA_F16_CASE(8)
            default: return ifa_fail(IFA_ERR_ARG, "ifa_gemv: unexpected nj %d", nj);
            }
#undef IFA_F16_CASE
        } else {
            IFA_REQUIRE(cols % 8 == 0 || true, "unreachable");
            k_gemv_f16w<0><<<dim3(grid), dim3(256), 0, s>>>(Wp, (int)rows, (int)cols, xp, bp, yp);
        }
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    size_t nblk = cols / (size_t)cap;
    IFA_DISPATCH_QUANT_DTYPE(w_dtype, k_gemv_f16x_quant<DT><<<dim3(grid), dim3(256), 0, s>>>((const uint8_t *)W, (int)rows, (int)nblk, xp, bp, yp));
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_gemv_tiled(int w_dtype, const void *Wt, size_t rows, size_t cols, const void *x_q8, const void *bias_f16,
                   void *y_f16, ifa_stream stream)
{
    IFA_REQUIRE(Wt && x_q8 && y_f16, "ifa_gemv_tiled: null pointer");
    IFA_REQUIRE(ax8_eligible(w_dtype), "ifa_gemv_tiled: dtype %d has no tiled layout", w_dtype);
    int cap = block_capacity(w_dtype);
    if (rows == 0) return IFA_OK;
    IFA_REQUIRE(cols > 0 && cols % (size_t)cap == 0, "ifa_gemv_tiled: cols %zu not a multiple of %d", cols, cap);
    return launch_ax8<true>(w_dtype, Wt, rows, cols, x_q8, bias_f16, y_f16, stream);
}

} // extern "C"
