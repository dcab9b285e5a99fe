// persistent decode layer kernel (ifa_decode_persist.h): Q3H_B64T1 instantiations (64-weight blocks: dim 4096 = 1 block
// per lane, ffn 11008 = 3)
#ifndef IFA_PS_SHAPES
#define IFA_PS_SHAPES(X) X(1, 1, 64) X(1, 3, 128) X(1, 4, 128)
#endif
#include "ifa_decode_persist_impl.h"

namespace ifa {
template int dec_persist_launch_dt<Q3H_B64T1>(int, int, int, int, const PsParams &, int, size_t, hipStream_t);
template bool dec_persist_has_dt<Q3H_B64T1>(int, int, int);
}
