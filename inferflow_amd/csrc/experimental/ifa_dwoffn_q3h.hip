// Wo rows in front of the W1 / W3 launch for Q3H_B64T1 weights (see ifa_decode_wo_ffn.h)
#include "ifa_dwoffn_impl.h"

namespace ifa {
template int dec_wo_ffn_launch_dt<Q3H_B64T1>(bool, const DecGemvParams &, const DecGemvParams &, const DecWoFfnExtra &, int, hipStream_t);
} // namespace ifa
