// ifa_decode_gemv_impl.h -- included by exactly one ifa_dgemv_<format>.hip per format.
#pragma once
#include <algorithm>
#include "ifa_decode_gemv.h"

namespace ifa {

// Tuning overrides (tools/sweep_variants.py builds one library per setting): 0 = the table below
#ifndef IFA_T_TH_QKV
#define IFA_T_TH_QKV 0
#endif
#ifndef IFA_T_TH_WO
#define IFA_T_TH_WO 0
#endif
#ifndef IFA_T_TH_GLU
#define IFA_T_TH_GLU 0
#endif
#ifndef IFA_T_TH_W2
#define IFA_T_TH_W2 0
#endif
#ifndef IFA_T_RW_QKV
#define IFA_T_RW_QKV 0
#endif
#ifndef IFA_T_RW_WO
#define IFA_T_RW_WO 0
#endif
#ifndef IFA_T_RW_GLU
#define IFA_T_RW_GLU 0
#endif
#ifndef IFA_T_RW_W2
#define IFA_T_RW_W2 0
#endif

#ifndef IFA_T_NP_QKV
#define IFA_T_NP_QKV -1
#endif
#ifndef IFA_T_NP_GLU
#define IFA_T_NP_GLU -1
#endif
#ifndef IFA_T_NP_W2
#define IFA_T_NP_W2 -1
#endif

// which of the four per-layer kernels a template instance is (by epilogue / prologue), for the overrides above
constexpr int dec_role(int epi, int norm) { return epi == EPI_GLU ? 2 : (epi == EPI_RESIDUAL ? (norm == 2 ? 1 : 3) : (epi == EPI_PLAIN && norm == 1 ? 0 : -1)); }

// rows (EPI_GLU: row pairs) a wave keeps in flight: bounded by registers, NM * RW * NJ * DW VGPRs
template <int DT>
constexpr int dec_rw(int epi, int norm, int nj, int nm, int th = DEC_THREADS)
{
    if (DT == Q4_B32T1A) {      // tuned on Llama-2-7B shapes (DESIGN.md "Kernel timeline")
        const int role = dec_role(epi, norm);
        const int forced = role == 0 ? IFA_T_RW_QKV : role == 1 ? IFA_T_RW_WO : role == 2 ? IFA_T_RW_GLU : role == 3 ? IFA_T_RW_W2 : 0;
        if (forced > 0) return forced;
        if (norm == 2 && nj == 2 && nm == 1) return th > DEC_THREADS ? 1 : 2;    // Wo without a prologue: 16 rows per CU, all requested at once (sweep r02)
        constexpr int a[9] = {0, 6, 6, 4, 4, 2, 2, 2, 2}, b[9] = {0, 6, 6, 3, 2, 2, 1, 1, 1};
        const int rw = nm == 2 ? b[nj] : a[nj];
        return th > DEC_THREADS ? (rw + 1) / 2 : rw;       // 1024 threads: 128 registers per lane, half the rows per wave
    }
    {   // (the sweep overrides apply to whichever format's translation unit is compiled with them)
        const int role = dec_role(epi, norm);
        const int forced = role == 0 ? IFA_T_RW_QKV : role == 1 ? IFA_T_RW_WO : role == 2 ? IFA_T_RW_GLU : role == 3 ? IFA_T_RW_W2 : 0;
        if (forced > 0) return forced;
    }
    // B64 formats (one block per lane at 4096 columns): sweep on Llama-2-7B Q3H + Q8 KV (configs[2];
    // profiles/r02_sweep_q3h.log): 544 -> 601 tok/s; the same entries measured on Q4_B64T1 / Q5_B64T1 (Q6_B64T1: 1024 threads spill)
    if (DT == Q3H_B64T1 || DT == Q3H_NATIVE || DT == Q4_B64T1 || DT == Q5_B64T1 || DT == Q6_B64T1) {
        if (DT != Q6_B64T1 && epi == EPI_GLU && nj == 1 && th > DEC_THREADS) return 3;   // W1/W3 at 1024 threads: 16.7 -> 14.1 us (Q3H)
        if ((epi == EPI_RESIDUAL || epi == EPI_PLAIN) && norm == 2 && nj == 1) return 2; // Wo without a prologue: 6.3 -> 4.6 us
        if (epi == EPI_RESIDUAL && norm == 0 && nj == 3) return 2;                    // W2 (11008 columns): 10.4 -> 9.1 us
    }
    if (DT == Q8_B32T2) {       // two blocks per lane at 4096 columns, like Q4_B32T1
        if (epi == EPI_GLU && nj == 2 && th > DEC_THREADS) return 2;
        if ((epi == EPI_RESIDUAL || epi == EPI_PLAIN) && norm == 2 && nj == 2) return 2;
    }
    int rw = 72 / (nm * nj * DecFmt<DT, 1>::DW);
    rw = rw < 1 ? 1 : (rw > 6 ? 6 : rw);
    return th > DEC_THREADS ? (rw + 1) / 2 : rw;           // 1024 threads: 128 registers per lane
}

// Workgroup size by kernel shape, measured on Llama-2-7B Q4: the gated W1/W3 kernel and the short Wo kernel
// (2 blocks per lane) gain from four waves per SIMD (issue stalls overlap), QKV and the long-row W2 kernel lose
template <int DT>
constexpr int dec_threads(int epi, int norm, int nj, bool xadd)
{
    if (!xadd) {
        const int role = dec_role(epi, norm);
        const int forced = role == 0 ? IFA_T_TH_QKV : role == 1 ? IFA_T_TH_WO : role == 2 ? IFA_T_TH_GLU : role == 3 ? IFA_T_TH_W2 : 0;
        if (forced > 0) return forced;
    }
    if ((DT == Q3H_B64T1 || DT == Q3H_NATIVE || DT == Q4_B64T1 || DT == Q5_B64T1) && nj == 1 && !xadd && epi == EPI_GLU) return 1024;          // (see dec_rw)
    if (DT == Q8_B32T2 && nj == 2 && !xadd && epi == EPI_GLU) return 1024;
    if (DT == Q4_B32T1A && nj == 2 && !xadd && epi == EPI_PLAIN && norm == 1) return 1024;      // QKV with the wave-specialised prologue (r04 sweep: 8.8 -> 8.3 us)
    return (DT == Q4_B32T1A && nj == 2 && !xadd && (epi == EPI_GLU || (epi == EPI_RESIDUAL && norm == 0))) ? 1024 : DEC_THREADS;
}

// waves that run the norm / quantiser prologue while the others already stream their rows (k_dec_gemv's NP; 0 = every wave
// takes part in the prologue, the round 1-3 form).  Only the dense per-layer kernels with a prologue.
template <int DT>
constexpr int dec_np(int epi, int norm, int nj, bool xadd, int th)
{
    if (xadd || norm == 2) return 0;
    const int role = dec_role(epi, norm);
    const int forced = role == 0 ? IFA_T_NP_QKV : role == 2 ? IFA_T_NP_GLU : role == 3 ? IFA_T_NP_W2 : -1;
    if (forced >= 0) return forced < th / 64 ? forced : 0;
    // round-4 sweep on Llama-2-7B Q4 (gpurun_out sweeps, DESIGN.md section 3 "wave-specialised prologue"): half of the waves
    // run the prologue -- QKV 9.6 -> 8.2-8.5 us (1024 threads), W1/W3 13.1 -> 12.95, W2 9.05 -> 8.0-8.1 (512 threads);
    // a quarter or three quarters of the waves lose (the prologue's VALU time, or too few early requests)
    if (role == 0 || role == 2 || role == 3) return th / 128;
    return 0;
}

template <int DT, int EPI, int NORM, bool XADD = false>
static int dec_gemv_launch_en(const DecGemvParams &P, int wgs_per_cu_opt, hipStream_t s)
{
    constexpr int NM = epi_is_glu(EPI) ? 2 : 1;
    const int nj = (P.nblk + 63) / 64;
    if (nj < 1) return ifa_fail(IFA_ERR_ARG, "fused GEMV: no columns");
    if (nj > DecGemvLimits<DT>::MAXNJ) {
        // long rows (w2 / wo of 34B-70B models): chunked kernel, no norm prologue, no GLU pair
        if constexpr (NORM == 0 && !XADD && (EPI == EPI_PLAIN || EPI == EPI_RESIDUAL || EPI == EPI_MOE_ACC || EPI == EPI_MOE_LAST)) {
            constexpr int MJ = DecGemvLimits<DT>::MAXNJ;
            const int nchunk = (nj + MJ - 1) / MJ;
            const int njl = (nj + nchunk - 1) / nchunk;
            if (nchunk > 4 || P.cols > 32768) return ifa_fail(IFA_ERR_ARG, "fused GEMV: %d columns exceed the limit of dtype %d", P.cols, DT);
            const int per_cu_l = wgs_per_cu_opt > 0 ? wgs_per_cu_opt : 1;
            int wgs_l = std::min(dec_num_cus() * per_cu_l, (P.total_rows + DEC_WAVES - 1) / DEC_WAVES);
            if (wgs_l < 1) wgs_l = 1;
            const size_t smem_l = xlds_bytes(P.cols);
#define IFA_DGL(NJV) \
    case NJV: if constexpr (NJV <= MJ) { \
        constexpr int RWL = dec_rw<DT>(-1, 0, NJV, 1) >= 2 ? dec_rw<DT>(-1, 0, NJV, 1) / 2 : 1; \
        auto kern = k_dec_gemv_long<DT, NJV, RWL, EPI>; \
        if (smem_l > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_l)); \
        kern<<<dim3((unsigned)wgs_l), dim3(DEC_THREADS), smem_l, s>>>(P); } break;
            switch (njl) { IFA_DGL(1) IFA_DGL(2) IFA_DGL(3) IFA_DGL(4) IFA_DGL(5) IFA_DGL(6) IFA_DGL(7) IFA_DGL(8) }
#undef IFA_DGL
            IFA_LAUNCH_CHECK();
            return IFA_OK;
        } else {
            return ifa_fail(IFA_ERR_ARG, "fused GEMV: %d columns exceed the limit of dtype %d for a normalised / gated input", P.cols, DT);
        }
    }
    // exactly one workgroup per CU (a second one would queue its activation behind the first one's weights)
    const int per_cu = wgs_per_cu_opt > 0 ? wgs_per_cu_opt : 1;
    const int waves = dec_threads<DT>(EPI, NORM, nj, XADD) / 64;
    int wgs = std::min(dec_num_cus() * per_cu, (P.total_rows + waves - 1) / waves);
    if (wgs < 1) wgs = 1;
    const dim3 grid((unsigned)wgs);
    if (wgs > 0xFFFF || P.nblk > 0xFFFF) return ifa_fail(IFA_ERR_ARG, "fused GEMV: grid %d / %d blocks per row exceed the packed launch scalars", wgs, P.nblk);
    const size_t smem = xlds_bytes(P.cols);
    // the matrix pointer as a preloaded kernel argument when the launch has ONE matrix (pair) and no expert table (k_dec_gemv)
    const uint8_t *pw0 = (P.nsets == 1 && !epi_is_moe(EPI) && !P.w_table) ? P.W0[0] : nullptr;
#ifdef IFA_NO_PRELOAD_W            // (A / B: the row addresses from the argument block, as before round 5)
    pw0 = nullptr;
#endif
    // inputs that are normalised or feed an activation are [dim] vectors (dim <= 8192): <= 4 blocks per lane of a B32 format
    constexpr int NJCAP = (NORM == 1 || EPI == EPI_GLU || EPI == EPI_ACT || EPI == EPI_MOE_GLU || EPI == EPI_MOE_ACT) ? 4 : 8;
    if (nj > NJCAP) return ifa_fail(IFA_ERR_ARG, "fused GEMV: %d columns exceed the limit for a normalised / gated input", P.cols);
#define IFA_DG(NJV) \
    case NJV: if constexpr (NJV <= DecGemvLimits<DT>::MAXNJ && NJV <= NJCAP) { \
        constexpr int THV = dec_threads<DT>(EPI, NORM, NJV, XADD); \
        constexpr int NPV = dec_np<DT>(EPI, NORM, NJV, XADD, THV); \
        auto kern = k_dec_gemv<DT, NJV, dec_rw<DT>(EPI, NORM, NJV, NM, THV), EPI, NORM, XADD, THV, NPV>; \
        if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        kern<<<grid, dim3(THV), smem, s>>>(P.x, P.norm_w, P.norm_b, P.cols, pw0, P.W1, (int)((unsigned)P.nblk | ((unsigned)wgs << 16)), P.total_rows, P); } break;
    switch (nj) { IFA_DG(1) IFA_DG(2) IFA_DG(3) IFA_DG(4) IFA_DG(5) IFA_DG(6) IFA_DG(7) IFA_DG(8) }
#undef IFA_DG
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

template <int DT>
int dec_gemv_launch_dt(int epi, int norm, const DecGemvParams &P, int wgs_per_cu, hipStream_t s)
{
    if (P.x_add) {      // tensor-parallel seams: the sum of the layer input and the merged product is formed in the prologue
        if (epi == EPI_PLAIN && norm == 1) return dec_gemv_launch_en<DT, EPI_PLAIN, 1, true>(P, wgs_per_cu, s);
        if (epi == EPI_GLU && norm == 1) return dec_gemv_launch_en<DT, EPI_GLU, 1, true>(P, wgs_per_cu, s);
        if (epi == EPI_ACT && norm == 1) return dec_gemv_launch_en<DT, EPI_ACT, 1, true>(P, wgs_per_cu, s);
        return ifa_fail(IFA_ERR_ARG, "fused GEMV: no x_add kernel for epilogue %d / norm %d", epi, norm);
    }
    if (epi == EPI_PLAIN && norm == 1) return dec_gemv_launch_en<DT, EPI_PLAIN, 1>(P, wgs_per_cu, s);
    if (epi == EPI_PLAIN && norm == 0) return dec_gemv_launch_en<DT, EPI_PLAIN, 0>(P, wgs_per_cu, s);
    if (epi == EPI_RESIDUAL && norm == 0) return dec_gemv_launch_en<DT, EPI_RESIDUAL, 0>(P, wgs_per_cu, s);
    if (epi == EPI_RESIDUAL && norm == 2) return dec_gemv_launch_en<DT, EPI_RESIDUAL, 2>(P, wgs_per_cu, s);
    if (epi == EPI_PLAIN && norm == 2) return dec_gemv_launch_en<DT, EPI_PLAIN, 2>(P, wgs_per_cu, s);
    if (epi == EPI_GLU && norm == 1) return dec_gemv_launch_en<DT, EPI_GLU, 1>(P, wgs_per_cu, s);
    if (epi == EPI_ACT && norm == 1) return dec_gemv_launch_en<DT, EPI_ACT, 1>(P, wgs_per_cu, s);
    if (epi == EPI_GLU && norm == 0) return dec_gemv_launch_en<DT, EPI_GLU, 0>(P, wgs_per_cu, s);
    if (epi == EPI_ACT && norm == 0) return dec_gemv_launch_en<DT, EPI_ACT, 0>(P, wgs_per_cu, s);
    if (epi == EPI_MOE_GLU && norm == 1) return dec_gemv_launch_en<DT, EPI_MOE_GLU, 1>(P, wgs_per_cu, s);
    if (epi == EPI_MOE_ACT && norm == 1) return dec_gemv_launch_en<DT, EPI_MOE_ACT, 1>(P, wgs_per_cu, s);
    if (epi == EPI_MOE_ACC && norm == 0) return dec_gemv_launch_en<DT, EPI_MOE_ACC, 0>(P, wgs_per_cu, s);
    if (epi == EPI_MOE_LAST && norm == 0) return dec_gemv_launch_en<DT, EPI_MOE_LAST, 0>(P, wgs_per_cu, s);
    return ifa_fail(IFA_ERR_ARG, "fused GEMV: no kernel for epilogue %d / norm %d", epi, norm);
}

} // namespace ifa
