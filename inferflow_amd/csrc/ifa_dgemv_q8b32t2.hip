// fused decode GEMV kernels for Q8_B32T2 weights (see ifa_decode_gemv.h)
#include "ifa_decode_gemv_impl.h"
namespace ifa { template int dec_gemv_launch_dt<Q8_B32T2>(int, int, const DecGemvParams &, int, hipStream_t); }
