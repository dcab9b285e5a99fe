// ifa_dqkvattn_impl.h -- included by exactly one ifa_dqkvattn_<format>.hip per weight format: the instantiations of
// k_dec_qkv_attn (ifa_decode_qkv_attn.h) for the shapes listed in dec_qkv_attn_shape().
#pragma once
#include "ifa_host.h"
#include "ifa_decode_qkv_attn.h"

namespace ifa {

template <int DT, int NJ, int RW, int NORM, bool Q8, int PB, bool KT, bool UL = false>
static int qa_go(const DecGemvParams &P, const DecAttnParams &A, const DecQkvAttnExtra &E, int max_ctx, hipStream_t s)
{
    constexpr int HD = 128;
    auto kern = k_dec_qkv_attn<DT, NJ, RW, NORM, HD, Q8, PB, KT, QA_THREADS / 128, UL>;
    const size_t smem = std::max(xlds_bytes(P.cols), dec_attn_smem(HD, max_ctx, KT ? PB : 0));
    if (smem > (size_t)160 * 1024) return ifa_fail(IFA_ERR_ARG, "fused QKV + attention: %zu bytes of LDS", smem);
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (A.heads > 255 || A.kv_heads > 255 || E.gk > 0xFFFF) return ifa_fail(IFA_ERR_ARG, "fused QKV + attention: geometry exceeds the packed launch scalar");
    const int pgeo = A.heads | (A.kv_heads << 8) | (E.gk << 16);
    kern<<<dim3((unsigned)(A.kv_heads * E.gk)), dim3(QA_THREADS), smem, s>>>(P.x, P.norm_w, P.norm_b, P.cols, pgeo, P.W0[0], P.W0[1], P.W0[2], P, A, E);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

template <int DT, int NJ, int RW, int NORM>
static int qa_pick(bool kv_q8, int pb, bool kt, const DecGemvParams &P, const DecAttnParams &A, const DecQkvAttnExtra &E, int max_ctx, hipStream_t s)
{
    if (kv_q8) {
        if (pb == 64) return qa_go<DT, NJ, RW, NORM, true, 64, false>(P, A, E, max_ctx, s);
        if (pb == 128) return qa_go<DT, NJ, RW, NORM, true, 128, false>(P, A, E, max_ctx, s);
        return qa_go<DT, NJ, RW, NORM, true, 256, false>(P, A, E, max_ctx, s);
    }
    if (kt) {
        if (pb == 64) return qa_go<DT, NJ, RW, NORM, false, 64, true>(P, A, E, max_ctx, s);
        if (pb == 128) return qa_go<DT, NJ, RW, NORM, false, 128, true>(P, A, E, max_ctx, s);
        return qa_go<DT, NJ, RW, NORM, false, 256, true>(P, A, E, max_ctx, s);
    }
    return qa_go<DT, NJ, RW, NORM, false, 256, false>(P, A, E, max_ctx, s);
}

// the UL kernels (heads' workgroups without rows): the 256-row bucket by default, the 128-row bucket as a measurement setting
template <int DT, int NJ, int RW, int NORM>
static int qa_pick_ul(bool kv_q8, int pb, bool kt, const DecGemvParams &P, const DecAttnParams &A, const DecQkvAttnExtra &E, int max_ctx, hipStream_t s)
{
    if (pb == 128) {
        if (kv_q8) return qa_go<DT, NJ, RW, NORM, true, 128, false, true>(P, A, E, max_ctx, s);
        if (kt) return qa_go<DT, NJ, RW, NORM, false, 128, true, true>(P, A, E, max_ctx, s);
    }
    if (kv_q8) return qa_go<DT, NJ, RW, NORM, true, 256, false, true>(P, A, E, max_ctx, s);
    if (kt) return qa_go<DT, NJ, RW, NORM, false, 256, true, true>(P, A, E, max_ctx, s);
    return qa_go<DT, NJ, RW, NORM, false, 256, false, true>(P, A, E, max_ctx, s);
}

// blocks per lane of a [dim]-column row of this format
template <int DT> constexpr int qa_nj(int cols) { return (cols / block_capacity(DT) + 63) / 64; }

template <int DT>
int dec_qkv_attn_launch_dt(int norm, bool kv_q8, int pb, bool kt, int rw, const DecGemvParams &P, const DecAttnParams &A, const DecQkvAttnExtra &E,
                           int max_ctx, hipStream_t s)
{
    constexpr int NJ4K = qa_nj<DT>(4096);       // dim 4096: Llama-2-7B (6 rows per wave), Mixtral-8x7B (3)
    const int nj = (P.nblk + 63) / 64;
    if (norm != 1) return ifa_fail(IFA_ERR_ARG, "fused QKV + attention: RMS-norm prologue only");
    // UL: (group + 2) * 128 rows over the gk - group workgroups without a head: 384 over 56 waves (7), 768 over 224 (4)
    if (E.unload && nj == NJ4K && rw == 6 && E.gk == 8) return qa_pick_ul<DT, NJ4K, 7, 1>(kv_q8, pb, kt, P, A, E, max_ctx, s);
    if (E.unload && nj == NJ4K && rw == 3 && E.gk == 32) return qa_pick_ul<DT, NJ4K, 4, 1>(kv_q8, pb, kt, P, A, E, max_ctx, s);
    if (nj == NJ4K && rw == 6) return qa_pick<DT, NJ4K, 6, 1>(kv_q8, pb, kt, P, A, E, max_ctx, s);
    if (nj == NJ4K && rw == 3) return qa_pick<DT, NJ4K, 3, 1>(kv_q8, pb, kt, P, A, E, max_ctx, s);
    return ifa_fail(IFA_ERR_ARG, "fused QKV + attention: no kernel for %d blocks per lane, %d rows per wave", nj, rw);
}

} // namespace ifa
