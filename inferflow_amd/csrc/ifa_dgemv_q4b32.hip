// fused decode GEMV kernels for Q4_B32T1A/B weights + the dtype dispatcher (see ifa_decode_gemv.h)
#include "ifa_decode_gemv_impl.h"

namespace ifa {

template int dec_gemv_launch_dt<Q4_B32T1A>(int, int, const DecGemvParams &, int, hipStream_t);

static int g_num_cus = 0;
int dec_num_cus()
{
    if (!g_num_cus) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

static int dec_maxnj(int dt)
{
    switch (dt) {
    case Q4_B32T1A: case Q4_B32T1B: return DecGemvLimits<Q4_B32T1A>::MAXNJ;
    case Q8_B32T2: return DecGemvLimits<Q8_B32T2>::MAXNJ;
    case Q4_B64T1: case Q3H_B64T1: case Q3H_NATIVE: case Q5_B64T1: case Q6_B64T1: return DecGemvLimits<Q4_B64T1>::MAXNJ;
    default: return 0;
    }
}

bool dec_gemv_supported(int w_dtype, size_t cols)
{
    const int mnj = dec_maxnj(w_dtype);
    if (mnj == 0 || cols == 0) return false;
    const size_t cap = (size_t)block_capacity(w_dtype);
    return cols % cap == 0 && cols / cap <= (size_t)(64 * mnj);
}

bool dec_gemv_supported_long(int w_dtype, size_t cols)
{
    const int mnj = dec_maxnj(w_dtype);
    if (mnj == 0 || cols == 0 || cols > 32768) return false;
    const size_t cap = (size_t)block_capacity(w_dtype);
    return cols % cap == 0 && cols / cap <= (size_t)(64 * mnj * 4);
}

int dec_gemv_launch(int w_dtype, int epi, int norm, const DecGemvParams &P0, int wgs_per_cu, hipStream_t s, long long *trace)
{
    DecGemvParams P = P0;
    P.trace = trace;
    P.total_rows = 0;
    for (int i = 0; i < P.nsets; i++) P.total_rows += P.rows[i];
    if (!dec_gemv_supported_long(w_dtype, (size_t)P.cols))
        return ifa_fail(IFA_ERR_ARG, "fused GEMV: dtype %d with %d columns is not supported", w_dtype, P.cols);
    P.nblk = P.cols / block_capacity(w_dtype);
    switch (w_dtype) {
    case Q4_B32T1A: case Q4_B32T1B: return dec_gemv_launch_dt<Q4_B32T1A>(epi, norm, P, wgs_per_cu, s);   // same bytes, same arithmetic
    case Q8_B32T2: return dec_gemv_launch_dt<Q8_B32T2>(epi, norm, P, wgs_per_cu, s);
    case Q4_B64T1: return dec_gemv_launch_dt<Q4_B64T1>(epi, norm, P, wgs_per_cu, s);
    case Q3H_B64T1: return dec_gemv_launch_dt<Q3H_B64T1>(epi, norm, P, wgs_per_cu, s);
    case Q3H_NATIVE: return dec_gemv_launch_dt<Q3H_NATIVE>(epi, norm, P, wgs_per_cu, s);
    case Q5_B64T1: return dec_gemv_launch_dt<Q5_B64T1>(epi, norm, P, wgs_per_cu, s);
    case Q6_B64T1: return dec_gemv_launch_dt<Q6_B64T1>(epi, norm, P, wgs_per_cu, s);
    default: return ifa_fail(IFA_ERR_DTYPE, "fused GEMV: dtype %d", w_dtype);
    }
}

} // namespace ifa
