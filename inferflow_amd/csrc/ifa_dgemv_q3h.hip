// fused decode GEMV kernels for Q3H_B64T1 weights (see ifa_decode_gemv.h)
#include "ifa_decode_gemv_impl.h"
namespace ifa { template int dec_gemv_launch_dt<Q3H_B64T1>(int, int, const DecGemvParams &, int, hipStream_t); }
