// Fused decode GEMV for weights that take fp16 activations: F16 tensors (every tensor below tensor_quant_threshold,
// src/transformer/network_builder.cc:1557-1562, and whole F16 models such as bin/llm_inference.tiny.ini) and the block
// formats the reference's int8 path does not cover (Q8_B32T1, Q5_B32T1, Q4_B16, Q3_B32T1A/B, Q2_B32T1A/B --
// GetUseFullQuantGemv, src/transformer/inference_worker.cc:2707-2730; their kernels: src/kernels/gemv.h:632-1497).
//
// Same launch interface, prologue (norm in the canonical order of ifa_math.h), row -> set mapping and epilogues as
// k_dec_gemv (ifa_decode_kernels.h), so a layer with such tensors keeps the 5-launch decode step.  The dot product of
// a row is the op-level kernel's (ifa_gemv.hip: k_gemv_f16w / k_gemv_f16x_quant): lane l takes chunk / block l, l+64, ...
// in ascending order as ONE fp32 fma chain, then the wave butterfly -- fused and op-by-op results are bit-identical.
// Weights are read in the reference byte layout (no tiled copy): Tensor::data.
#include "ifa_decode_gemv.h"
#include "ifa_codec.h"

namespace ifa {

constexpr int DH_ROWS = 2;       // rows (EPI_GLU: row pairs) per wave and pass

template <int DT>
struct HRow {
    const uint8_t *p;
    // F16: 4 chunks of 8 halfs in flight per lane and row
    __device__ __forceinline__ float dot(const half_t *__restrict__ xs, int cols, int nblk, int lane) const
    {
        float acc = 0.0f;
        if constexpr (DT == F16) {
            const int chunks = cols >> 3;
            const u32x4 *w = reinterpret_cast<const u32x4 *>(p);
            const u32x4 *xv = reinterpret_cast<const u32x4 *>(xs);
            for (int c0 = lane; c0 < chunks; c0 += 256) {
                u32x4 wr[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { const int c = c0 + 64 * j; wr[j] = w[min(c, chunks - 1)]; }
#pragma unroll
                for (int j = 0; j < 4; j++) { const int c = c0 + 64 * j; if (c < chunks) acc = dot8_f16(wr[j], xv[c], acc); }
            }
        } else {
            constexpr int CAP = block_capacity(DT), BB = block_bytes(DT);
            for (int blk = lane; blk < nblk; blk += 64) {
                RawBlock<BB> b;
                b.load(p + (size_t)blk * BB);
                int q[CAP]; float scale, base;
                decode_block<DT>(b, q, scale, base);
                const half_t *xp = xs + (size_t)blk * CAP;
#pragma unroll
                for (int i = 0; i < CAP; i++) {
                    const float wv = h2f(f2h(block_value<DT>(q[i], scale, base)));     // half-rounded like the reference's arr_a[]
                    acc = __builtin_fmaf(wv, h2f(xp[i]), acc);
                }
            }
        }
        return acc;
    }
};

template <int DT, int EPI, int NORM, bool XADD>
__global__ void __launch_bounds__(DEC_THREADS) k_dec_gemv_h(const DecGemvParams P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *xs = reinterpret_cast<half_t *>(smem);                                                   // [cols]
    float *part = reinterpret_cast<float *>(smem + (((size_t)P.cols * 2 + 15) & ~(size_t)15));     // [128] group sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunks = P.cols >> 3;
    const int nk = (chunks + DEC_THREADS - 1) / DEC_THREADS;
    // ---- stage the activation (XADD: the sum of the layer input and the merged product, TensorOpr::Add order)
    for (int k = 0; k < nk; k++) {
        const int c = tid + k * DEC_THREADS;
        half8_t v;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (half_t)0;
        if (c < chunks) {
            v = *reinterpret_cast<const half8_t *>(P.x + (size_t)c * 8);
            if constexpr (XADD) {
                half8_t a = *reinterpret_cast<const half8_t *>(P.x_add + (size_t)c * 8);
                if (P.x_add_bias) {
                    const half8_t ab = *reinterpret_cast<const half8_t *>(P.x_add_bias + (size_t)c * 8);
#pragma unroll
                    for (int i = 0; i < 8; i++) a[i] = f2h(h2f(a[i]) + h2f(ab[i]));
                }
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = f2h(h2f(v[i]) + h2f(a[i]));
                if (P.xsum_out && blockIdx.x == 0) *reinterpret_cast<half8_t *>(P.xsum_out + (size_t)c * 8) = v;
            }
            *reinterpret_cast<half8_t *>(xs + (size_t)c * 8) = v;
        }
        if constexpr (NORM == 1) {      // canonical order: chunk c is lane c % 64 of group c / 64
            const float pg = wave_sum(rms_chunk_sq(v));
            if (lane == 0) part[wave + k * DEC_WAVES] = pg;
        }
    }
    __syncthreads();
    if constexpr (NORM == 1) {
        const float scale = rms_scale_of(rms_total(part, (chunks + 63) >> 6), P.cols, P.eps);
        for (int k = 0; k < nk; k++) {
            const int c = tid + k * DEC_THREADS;
            if (c >= chunks) continue;
            const half8_t v = *reinterpret_cast<const half8_t *>(xs + (size_t)c * 8);
            half8_t o;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float t = (float)v[i] * scale;
                if (P.norm_w) {
                    const float mlt = P.multi_base + h2f(P.norm_w[(size_t)c * 8 + i]);
                    t = t * mlt;
                    if (P.norm_b) t = t + h2f(P.norm_b[(size_t)c * 8 + i]);
                }
                o[i] = f2h(t);
            }
            *reinterpret_cast<half8_t *>(xs + (size_t)c * 8) = o;
            if (P.xn_out && blockIdx.x == 0) *reinterpret_cast<half8_t *>(P.xn_out + (size_t)c * 8) = o;
        }
        __syncthreads();
    }
    // ---- rows, strided over the waves of the grid
    constexpr int NM = epi_is_glu(EPI) ? 2 : 1;
    const int gw = blockIdx.x * DEC_WAVES + wave;
    const int W = gridDim.x * DEC_WAVES;
    const size_t row_bytes = DT == F16 ? (size_t)P.cols * 2 : (size_t)P.nblk * block_bytes(DT);
    for (int v0 = gw; v0 < P.total_rows; v0 += DH_ROWS * W) {
        float a[NM][DH_ROWS];
#pragma unroll
        for (int i = 0; i < DH_ROWS; i++) {
            const int v = min(v0 + i * W, P.total_rows - 1);
            const DecRow d = dec_locate(P, v);
            HRow<DT> r0{d.W0 + (size_t)d.row * row_bytes};
            a[0][i] = r0.dot(xs, P.cols, P.nblk, lane);
            if constexpr (NM == 2) { HRow<DT> r1{d.W1 + (size_t)d.row * row_bytes}; a[1][i] = r1.dot(xs, P.cols, P.nblk, lane); }
        }
#pragma unroll
        for (int i = 0; i < DH_ROWS; i++)
#pragma unroll
            for (int mm = 0; mm < NM; mm++) a[mm][i] = wave_sum(a[mm][i]);
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int i = 0; i < DH_ROWS; i++)
            if (lane == i) { a0 = a[0][i]; if constexpr (NM == 2) a1 = a[1][i]; }
        const int v = v0 + lane * W;
        if (lane < DH_ROWS && v < P.total_rows) {
            const DecRow d = dec_locate(P, v);
            half_t res = (half_t)0, res2 = (half_t)0;
            if constexpr (EPI == EPI_RESIDUAL) { res = P.residual[d.row]; if (P.residual2) res2 = P.residual2[d.row]; }
            dec_finish_row<EPI>(P, d, a0, a1, res, res2);
        }
    }
}

bool dec_gemv_h_supported(int w_dtype, size_t cols)
{
    if (w_dtype == F32 || cols == 0 || cols > 32768 || cols % 8 != 0) return false;
    const int cap = block_capacity(w_dtype);
    return cap > 0 && cols % (size_t)cap == 0;
}

template <int DT, int EPI, int NORM, bool XADD>
static int launch_h(const DecGemvParams &P, hipStream_t s)
{
    const size_t smem = (((size_t)P.cols * 2 + 15) & ~(size_t)15) + 132 * 4;
    const int per_wg = DEC_WAVES * DH_ROWS;
    int wgs = std::min(dec_num_cus() * 2, (P.total_rows + per_wg - 1) / per_wg);
    if (wgs < 1) wgs = 1;
    auto kern = k_dec_gemv_h<DT, EPI, NORM, XADD>;
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)wgs), dim3(DEC_THREADS), smem, s>>>(P);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

template <int DT>
static int launch_h_dt(int epi, int norm, const DecGemvParams &P, hipStream_t s)
{
    if (P.x_add) {
        if (epi == EPI_PLAIN && norm == 1) return launch_h<DT, EPI_PLAIN, 1, true>(P, s);
        if (epi == EPI_GLU && norm == 1) return launch_h<DT, EPI_GLU, 1, true>(P, s);
        if (epi == EPI_ACT && norm == 1) return launch_h<DT, EPI_ACT, 1, true>(P, s);
        return ifa_fail(IFA_ERR_ARG, "fused fp16-activation GEMV: no x_add kernel for epilogue %d / norm %d", epi, norm);
    }
    if (epi == EPI_PLAIN && norm == 1) return launch_h<DT, EPI_PLAIN, 1, false>(P, s);
    if (epi == EPI_PLAIN && norm == 0) return launch_h<DT, EPI_PLAIN, 0, false>(P, s);
    if (epi == EPI_RESIDUAL && norm == 0) return launch_h<DT, EPI_RESIDUAL, 0, false>(P, s);
    if (epi == EPI_GLU && norm == 1) return launch_h<DT, EPI_GLU, 1, false>(P, s);
    if (epi == EPI_ACT && norm == 1) return launch_h<DT, EPI_ACT, 1, false>(P, s);
    if (epi == EPI_GLU && norm == 0) return launch_h<DT, EPI_GLU, 0, false>(P, s);
    if (epi == EPI_ACT && norm == 0) return launch_h<DT, EPI_ACT, 0, false>(P, s);
    return ifa_fail(IFA_ERR_ARG, "fused fp16-activation GEMV: no kernel for epilogue %d / norm %d", epi, norm);
}

// epi / norm as dec_gemv_launch; P.W0 / P.W1 point at the reference-layout bytes (Tensor::data), P.x at F16 values
int dec_gemv_h_launch(int w_dtype, int epi, int norm, const DecGemvParams &P0, hipStream_t s)
{
    DecGemvParams P = P0;
    P.trace = nullptr;
    P.total_rows = 0;
    for (int i = 0; i < P.nsets; i++) P.total_rows += P.rows[i];
    if (!dec_gemv_h_supported(w_dtype, (size_t)P.cols))
        return ifa_fail(IFA_ERR_ARG, "fused fp16-activation GEMV: dtype %d with %d columns is not supported", w_dtype, P.cols);
    P.nblk = P.cols / block_capacity(w_dtype);
    if (w_dtype == F16) return launch_h_dt<F16>(epi, norm, P, s);
    IFA_DISPATCH_QUANT_DTYPE(w_dtype, return launch_h_dt<DT>(epi, norm, P, s));
    return IFA_OK;
}

} // namespace ifa
