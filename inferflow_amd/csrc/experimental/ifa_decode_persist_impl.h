// ifa_decode_persist_impl.h -- included by exactly one ifa_dpersist_<format>.hip per weight format: the shapes of
// k_dec_persist (ifa_decode_persist.h) that exist for the format, and their launcher.
#pragma once
#include "ifa_decode_persist_launch.h"

namespace ifa {

template <int DT, int NJA, int NJB, int HD, bool Q8>
static int ps_launch_one(const PsParams &P, int ncu, size_t smem, hipStream_t s)
{
    auto kern = k_dec_persist<DT, NJA, NJB, HD, Q8>;
    IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)ncu), dim3(PS_THREADS), smem, s>>>(P);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// (blocks per lane of a dim-wide row, of an ffn-wide row, head size) instantiated per format: Llama-2-7B / 13B,
// Mistral-7B-like shapes and the small test models; anything else keeps the five-launch layer
template <int DT> struct PsShapes;

template <int DT>
int dec_persist_launch_dt(int nja, int njb, int hd, int kvq8, const PsParams &P, int ncu, size_t smem, hipStream_t s)
{
#define IFA_PS(A, B, H) \
    if (nja == A && njb == B && hd == H) return kvq8 ? ps_launch_one<DT, A, B, H, true>(P, ncu, smem, s) : ps_launch_one<DT, A, B, H, false>(P, ncu, smem, s);
    IFA_PS_SHAPES(IFA_PS)
#undef IFA_PS
    return ifa_fail(IFA_ERR_ARG, "persistent decode: no kernel for dtype %d with %d / %d blocks per lane, head_dim %d", DT, nja, njb, hd);
}

template <int DT>
bool dec_persist_has_dt(int nja, int njb, int hd)
{
#define IFA_PS(A, B, H) if (nja == A && njb == B && hd == H) return true;
    IFA_PS_SHAPES(IFA_PS)
#undef IFA_PS
    return false;
}

} // namespace ifa
