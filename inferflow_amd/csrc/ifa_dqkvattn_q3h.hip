// fused QKV + attention launch for Q3H_B64T1 weights (see ifa_decode_qkv_attn.h)
#include <algorithm>
#include "ifa_dqkvattn_impl.h"

namespace ifa {
template int dec_qkv_attn_launch_dt<Q3H_B64T1>(int, bool, int, bool, int, const DecGemvParams &, const DecAttnParams &, const DecQkvAttnExtra &, int, hipStream_t);
} // namespace ifa
