// persistent decode layer kernel (ifa_decode_persist.h): Q4_B32T1A/B instantiations.  Shapes = (blocks per lane of a
// dim-wide row, of an ffn-wide row, head size): the small test models, Llama-2-7B, Mistral-7B-like, Llama-2-13B
#ifndef IFA_PS_SHAPES
#define IFA_PS_SHAPES(X) X(1, 1, 64) X(2, 6, 128) X(2, 7, 128) X(3, 7, 128)
#endif
#include "ifa_decode_persist_impl.h"

namespace ifa {
template int dec_persist_launch_dt<Q4_B32T1A>(int, int, int, int, const PsParams &, int, size_t, hipStream_t);
template bool dec_persist_has_dt<Q4_B32T1A>(int, int, int);
}
