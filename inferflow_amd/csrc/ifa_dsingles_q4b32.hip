// the single-row experts of a batched MoE step on Q4_B32T1A / B expert tables (ifa_decode_singles.h): instances + launcher
#include <algorithm>
#include "ifa_host.h"
#include "ifa_decode_gemv.h"
#include "ifa_decode_singles.h"

namespace ifa {

bool dec_singles_supported(int w_dtype, size_t rows, size_t cols, bool glu)
{
    if (w_dtype != Q4_B32T1A && w_dtype != Q4_B32T1B) return false;
    if (rows == 0 || cols % 32 != 0 || cols == 0 || cols > (glu ? 8192u : 16384u)) return false;
    return xlds_bytes((int)cols) <= (size_t)96 * 1024;
}

template <int NJ, int RW, bool GLU>
static int ds_launch(const DecSinglesParams &S, int wgs, int max_singles, hipStream_t s)
{
    auto kern = k_dec_singles<Q4_B32T1A, NJ, RW, GLU>;
    const size_t smem = xlds_bytes(S.cols);
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)wgs, (unsigned)max_singles), dim3(DEC_THREADS), smem, s>>>(S);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int dec_singles_launch(int w_dtype, const DecSinglesParams &S, bool glu, int max_singles, hipStream_t s)
{
    if (!dec_singles_supported(w_dtype, (size_t)S.rows, (size_t)S.cols, glu)) return ifa_fail(IFA_ERR_ARG, "single-row experts: dtype %d, %d x %d", w_dtype, S.rows, S.cols);
    if (max_singles <= 0) return IFA_OK;
    const int nj = (S.nblk + 63) / 64;
    // every present single gets the whole chip (one workgroup per CU); the slots run one after the other as workgroups retire
    const int wgs = std::max(1, std::min(dec_num_cus(), (S.rows + DEC_WAVES - 1) / DEC_WAVES));
    // rows (pairs) in flight per wave: NM x RW x NJ x 5 registers
    if (glu) {
        switch (nj) {
        case 1: return ds_launch<1, 6, true>(S, wgs, max_singles, s);
        case 2: return ds_launch<2, 4, true>(S, wgs, max_singles, s);
        case 3: return ds_launch<3, 3, true>(S, wgs, max_singles, s);
        default: return ds_launch<4, 2, true>(S, wgs, max_singles, s);
        }
    }
    switch (nj) {
    case 1: return ds_launch<1, 6, false>(S, wgs, max_singles, s);
    case 2: return ds_launch<2, 6, false>(S, wgs, max_singles, s);
    case 3: return ds_launch<3, 4, false>(S, wgs, max_singles, s);
    case 4: return ds_launch<4, 4, false>(S, wgs, max_singles, s);
    case 5: return ds_launch<5, 2, false>(S, wgs, max_singles, s);
    case 6: return ds_launch<6, 2, false>(S, wgs, max_singles, s);
    case 7: return ds_launch<7, 2, false>(S, wgs, max_singles, s);
    default: return ds_launch<8, 2, false>(S, wgs, max_singles, s);
    }
}

} // namespace ifa
