// ifa_gemm_rows_mo.hip -- the 2..16-row weight-streaming GEMM (ifa_gemm_rows_mfma.hip) on weights in the MO layout
// ("MFMA operand order", ifa_gemm_rows_mfma.h): the instantiations and their launcher.  A lane's 16-byte request is its own
// A operand, a superstep of a tile one contiguous KiB per wave; 16 activation rows x 4096 columns fit the LDS (no patches), so
// wq | wk | wv, wo and w1 / w3 of a 4096-wide model are single-chunk kernels (CH = 1: straight-line, counted waits) for every
// batch of 2..16 rows; longer rows (w2) walk 4096-column chunks.  Rows are staged as 8 or 16 (TX).
#include "ifa_gemm_rows_mfma_body.h"

namespace ifa {

template <int MT, int TX, int EPI, int NORM, int CH>
static int mo_launch4(int wgs, size_t smem, const GmArgs &P, hipStream_t s)
{
    auto kern = k_gemm_rows_mfma<MT, TX, EPI, NORM, true, CH>;
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)wgs), dim3(GM_THREADS), smem, s>>>(P);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

template <int MT, int TX>
static int mo_launch2(int wgs, size_t smem, const GmArgs &P, int epi, int norm, bool one, hipStream_t s)
{
    if (one) {
        if (epi == GM_PLAIN && norm == 0) return mo_launch4<MT, TX, GM_PLAIN, 0, 1>(wgs, smem, P, s);
        if (epi == GM_RESIDUAL && norm == 0) return mo_launch4<MT, TX, GM_RESIDUAL, 0, 1>(wgs, smem, P, s);
        if (epi == GM_PLAIN && norm == 1) return mo_launch4<MT, TX, GM_PLAIN, 1, 1>(wgs, smem, P, s);
        if constexpr (MT % 2 == 0) {
            if (epi == GM_GLU && norm == 1) return mo_launch4<MT, TX, GM_GLU, 1, 1>(wgs, smem, P, s);
            if (epi == GM_GLU && norm == 0) return mo_launch4<MT, TX, GM_GLU, 0, 1>(wgs, smem, P, s);
        }
    } else {
        if (epi == GM_PLAIN && norm == 0) return mo_launch4<MT, TX, GM_PLAIN, 0, 0>(wgs, smem, P, s);
        if (epi == GM_RESIDUAL && norm == 0) return mo_launch4<MT, TX, GM_RESIDUAL, 0, 0>(wgs, smem, P, s);
        if constexpr (MT % 2 == 0) { if (epi == GM_GLU && norm == 0) return mo_launch4<MT, TX, GM_GLU, 0, 0>(wgs, smem, P, s); }
    }
    return ifa_fail(IFA_ERR_ARG, "rows GEMM (MO): no kernel for epilogue %d / norm %d / %s", epi, norm, one ? "one chunk" : "chunked");
}

// 17..32 rows: two column tiles per A operand, rows staged in 2048-column chunks (always the chunk loop, no norm prologue)
template <int MT>
static int mo_launch32(int wgs, size_t smem, const GmArgs &P, int epi, int norm, hipStream_t s)
{
    if (epi == GM_PLAIN && norm == 0) return mo_launch4<MT, 32, GM_PLAIN, 0, 0>(wgs, smem, P, s);
    if (epi == GM_RESIDUAL && norm == 0) return mo_launch4<MT, 32, GM_RESIDUAL, 0, 0>(wgs, smem, P, s);
    if constexpr (MT % 2 == 0) { if (epi == GM_GLU && norm == 0) return mo_launch4<MT, 32, GM_GLU, 0, 0>(wgs, smem, P, s); }
    return ifa_fail(IFA_ERR_ARG, "rows GEMM (MO, 17..32 rows): no kernel for epilogue %d / norm %d", epi, norm);
}

template <int MT>
static int mo_launch1(int wgs, size_t smem, const GmArgs &P, int epi, int norm, bool one, hipStream_t s)
{
    if (P.T <= 8) return mo_launch2<MT, 8>(wgs, smem, P, epi, norm, one, s);
    if (P.T <= 16) return mo_launch2<MT, 16>(wgs, smem, P, epi, norm, one, s);
    return mo_launch32<MT>(wgs, smem, P, epi, norm, s);
}

int gemm_rows_mo_launch(const GmArgs &P, int epi, int norm, int wgs, int maxt, hipStream_t s)
{
    const bool one = P.nblk * 32 <= GmGeo<32>::CHUNK_COLS && P.T <= 16;
    if (norm == 1 && !one) return ifa_fail(IFA_ERR_ARG, "rows GEMM (MO): the norm prologue needs the whole row in one chunk");
    const size_t smem = gm_smem(P.T, maxt, 1);
    switch (maxt) {
    case 1: return mo_launch1<1>(wgs, smem, P, epi, norm, one, s);
    case 2: return mo_launch1<2>(wgs, smem, P, epi, norm, one, s);
    case 3: return mo_launch1<3>(wgs, smem, P, epi, norm, one, s);
    case 4: return mo_launch1<4>(wgs, smem, P, epi, norm, one, s);
    case 6: return mo_launch1<6>(wgs, smem, P, epi, norm, one, s);
    default: return mo_launch1<8>(wgs, smem, P, epi, norm, one, s);
    }
}

} // namespace ifa
