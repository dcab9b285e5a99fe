// ifa_gemm_rows_gemv.hip -- instantiations and launcher of the 2..4-row streaming kernel (ifa_gemm_rows_gemv.h)
#include <algorithm>
#include "ifa_host.h"
#include "ifa_gemm_rows_gemv.h"

namespace ifa {

static int gv_cus()
{
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t prop; ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256; }
    return ncu;
}

bool gemm_rows_gemv_ok(const GmArgs &P, int epi, int norm)
{
    if (P.mo || P.T < 2 || P.T > 4 || P.nblk < 1 || P.nblk > 512) return false;
    if (epi != GM_PLAIN && epi != GM_RESIDUAL && epi != GM_GLU) return false;
    if (norm != 0 && norm != 1) return false;
    if (epi == GM_GLU && P.nsets != 1) return false;
    const size_t smem = (size_t)P.T * ((P.nblk + 63) / 64) * 4096 + (size_t)P.T * ((P.nblk * 4 + 63) / 64) * 4 + 16;
    return smem <= (size_t)160 * 1024 && !(norm == 1 && epi == GM_RESIDUAL);
}

template <int NJ, int RW, int EPI, int NORM, int T>
static int gv_launch5(const GmArgs &P, int wgs, size_t smem, hipStream_t s)
{
    auto kern = k_rows_gemv<NJ, RW, EPI, NORM, T>;
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)wgs), dim3(GV_THREADS), smem, s>>>(P);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

template <int NJ, int RW, int EPI, int NORM>
static int gv_launch4(const GmArgs &P, int wgs, size_t smem, hipStream_t s)
{
    switch (P.T) {
    case 2: return gv_launch5<NJ, RW, EPI, NORM, 2>(P, wgs, smem, s);
    case 3: return gv_launch5<NJ, RW, EPI, NORM, 3>(P, wgs, smem, s);
    default: return gv_launch5<NJ, RW, EPI, NORM, 4>(P, wgs, smem, s);
    }
}

// RW by NJ: the rows in flight of a wave cost RW x NJ x 5 VGPRs twice (current + next batch), GLU twice that again
template <int NJ>
static int gv_launch3(const GmArgs &P, int epi, int norm, int wgs, size_t smem, hipStream_t s)
{
    constexpr int RWP = NJ <= 2 ? 3 : (NJ <= 4 ? 2 : 1);          // plain / residual
    constexpr int RWG = NJ <= 2 ? 2 : 1;                          // GLU: row pairs
    if (epi == GM_PLAIN) return norm ? gv_launch4<NJ, RWP, GM_PLAIN, 1>(P, wgs, smem, s) : gv_launch4<NJ, RWP, GM_PLAIN, 0>(P, wgs, smem, s);
    if (epi == GM_RESIDUAL) return gv_launch4<NJ, RWP, GM_RESIDUAL, 0>(P, wgs, smem, s);
    return norm ? gv_launch4<NJ, RWG, GM_GLU, 1>(P, wgs, smem, s) : gv_launch4<NJ, RWG, GM_GLU, 0>(P, wgs, smem, s);
}

int gemm_rows_gemv_launch(const GmArgs &P0, int epi, int norm, hipStream_t s)
{
    if (!gemm_rows_gemv_ok(P0, epi, norm)) return ifa_fail(IFA_ERR_ARG, "rows GEMV: T %d, nblk %d, epilogue %d, norm %d", P0.T, P0.nblk, epi, norm);
    GmArgs P = P0;
    P.total_rows = P.rows[0] + (P.nsets > 1 ? P.rows[1] : 0) + (P.nsets > 2 ? P.rows[2] : 0);
    const int nj = (P.nblk + 63) / 64;
    const size_t smem = (size_t)P.T * ((P.nblk + 63) / 64) * 4096 + (size_t)P.T * ((P.nblk * 4 + 63) / 64) * 4 + 16;
    const int rw = epi == GM_GLU ? (nj <= 2 ? 2 : 1) : (nj <= 2 ? 3 : (nj <= 4 ? 2 : 1));
    const int nbatch = (P.total_rows + rw - 1) / rw;
    const int per_cu = smem <= (size_t)72 * 1024 ? 2 : 1;         // two workgroups per CU when their row images fit the LDS side by side
    const int wgs = std::max(1, std::min(gv_cus() * per_cu, (nbatch + GV_WAVES - 1) / GV_WAVES));
    switch (nj) {
    case 1: return gv_launch3<1>(P, epi, norm, wgs, smem, s);
    case 2: return gv_launch3<2>(P, epi, norm, wgs, smem, s);
    case 3: return gv_launch3<3>(P, epi, norm, wgs, smem, s);
    case 4: return gv_launch3<4>(P, epi, norm, wgs, smem, s);
    case 5: return gv_launch3<5>(P, epi, norm, wgs, smem, s);
    case 6: return gv_launch3<6>(P, epi, norm, wgs, smem, s);
    case 7: return gv_launch3<7>(P, epi, norm, wgs, smem, s);
    default: return gv_launch3<8>(P, epi, norm, wgs, smem, s);
    }
}

} // namespace ifa
