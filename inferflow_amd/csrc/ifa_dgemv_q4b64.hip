// fused decode GEMV kernels for Q4_B64T1 weights (see ifa_decode_gemv.h)
#include "ifa_decode_gemv_impl.h"
namespace ifa { template int dec_gemv_launch_dt<Q4_B64T1>(int, int, const DecGemvParams &, int, hipStream_t); }
