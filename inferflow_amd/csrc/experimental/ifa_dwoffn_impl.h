// ifa_dwoffn_impl.h -- included by exactly one ifa_dwoffn_<format>.hip per weight format: the instantiations of k_dec_wo_ffn
// (ifa_decode_wo_ffn.h) for 4096-column rows (Llama-2-7B widths: 3 W1 / W3 row pairs per loader wave, <= 2 per front wave, 4 Wo rows per front wave).
#pragma once
#include <algorithm>
#include "ifa_host.h"
#include "ifa_decode_wo_ffn.h"

namespace ifa {

template <int DT> constexpr int wf_nj(int cols) { return (cols / block_capacity(DT) + 63) / 64; }

template <int DT>
int dec_wo_ffn_launch_dt(bool glu, const DecGemvParams &PW, const DecGemvParams &P, const DecWoFfnExtra &E, int num_cus, hipStream_t s)
{
    constexpr int NJ = wf_nj<DT>(4096);
    const size_t smem = xlds_bytes(P.cols);
    const dim3 grid((unsigned)num_cus), block(WF_THREADS);
    if (glu) k_dec_wo_ffn<DT, NJ, 3, 2, 4, EPI_GLU><<<grid, block, smem, s>>>(PW.x, P.norm_w, P.norm_b, P.cols, PW, P, E);
    else k_dec_wo_ffn<DT, NJ, 3, 2, 4, EPI_ACT><<<grid, block, smem, s>>>(PW.x, P.norm_w, P.norm_b, P.cols, PW, P, E);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

} // namespace ifa
