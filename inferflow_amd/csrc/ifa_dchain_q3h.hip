// chained decode launches (ifa_decode_chain.h) for Q3H_B64T1 weights (nibble-pair tiled form)
#include "ifa_dchain_impl.h"
namespace ifa {
template int dec_chain_launch_dt<Q3H_B64T1>(bool, int, bool, const DecGemvParams &, const DecGemvParams &, const DecGemvParams &, const DecChainExtra &, int, hipStream_t);
}
