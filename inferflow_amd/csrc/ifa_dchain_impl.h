// ifa_dchain_impl.h -- included by exactly one ifa_dchain_<format>.hip per format (launcher of k_dec_chain, ifa_decode_chain.h).
#pragma once
#include <algorithm>
#include "ifa_host.h"
#include "ifa_decode_chain.h"

namespace ifa {

constexpr int CHAIN_TH = 1024;

// W1 (| W3) rows per wave and pass: what 128 registers per lane hold next to the activation image
template <int DT> constexpr int chain_rw(int nja) { return (DT == Q4_B32T1A) ? (nja <= 2 ? 3 : 2) : (nja <= 1 ? 3 : 1); }

template <int DT, int NJA, int NJB, bool WO, int RO, int R2>
static int chain_run(const DecGemvParams &P, const DecGemvParams &Q, const DecGemvParams &PW, const DecChainExtra &E, int grid, hipStream_t s)
{
    constexpr int RW = chain_rw<DT>(NJA);
    auto kern = k_dec_chain<DT, NJA, NJB, RW, EPI_GLU, 1, WO, RO, R2, CHAIN_TH>;
    const size_t smem = (xlds_bytes(P.cols) + 15) / 16 * 16 + (xlds_bytes(Q.cols) + 15) / 16 * 16 + 64;      // two images + control words
    if (!wait_grid_fits((const void *)kern, CHAIN_TH, smem, grid))
        return ifa_fail(IFA_ERR_STATE, "chained FFN launch: %d workgroups cannot be resident at once on this device", grid);
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)grid), dim3(CHAIN_TH), smem, s>>>(WO ? PW.x : P.x, P.norm_w, P.norm_b, P.cols, P.W0[0], P.W1,
                                                            (int)((unsigned)P.nblk | ((unsigned)grid << 16)), P.total_rows, P, Q, PW, E);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

template <int DT>
int dec_chain_launch_dt(bool glu, int norm, bool wo, const DecGemvParams &P, const DecGemvParams &Q, const DecGemvParams &PW,
                        const DecChainExtra &E, int num_cus, hipStream_t s)
{
    if (!glu || norm != 1) return ifa_fail(IFA_ERR_ARG, "chained FFN launch: gated FFN behind an RMS norm only");
    const int nja = (P.nblk + 63) / 64, njb = (Q.nblk + 63) / 64;
    const int grid = num_cus, W = grid * (CHAIN_TH / 64);
    const int WL = grid * (CHAIN_TH / 128);                     // loader waves (half of each workgroup) take the W2 rows
    const int r2 = (Q.total_rows + WL - 1) / WL, ro = wo ? (PW.total_rows + W - 1) / W : 1;
    if (r2 < 1 || r2 > 2 || ro != 1) return ifa_fail(IFA_ERR_ARG, "chained FFN launch: %d output rows over %d waves", Q.total_rows, W);
#define IFA_CH(A, B) \
    if (nja == A && njb == B) { \
        if (wo) return r2 == 1 ? chain_run<DT, A, B, true, 1, 1>(P, Q, PW, E, grid, s) : chain_run<DT, A, B, true, 1, 2>(P, Q, PW, E, grid, s); \
        return r2 == 1 ? chain_run<DT, A, B, false, 1, 1>(P, Q, PW, E, grid, s) : chain_run<DT, A, B, false, 1, 2>(P, Q, PW, E, grid, s); }
    if constexpr (DT == Q4_B32T1A) { IFA_CH(2, 6) IFA_CH(2, 7) }
    else { IFA_CH(1, 3) IFA_CH(1, 4) }
#undef IFA_CH
    return ifa_fail(IFA_ERR_ARG, "chained FFN launch: no instance for %d / %d blocks per lane", nja, njb);
}

} // namespace ifa
