// persistent decode layer kernel (ifa_decode_persist.h): weight-format dispatcher.  The kernels of each format are
// instantiated in their own translation unit (ifa_dpersist_<format>.hip) so that the formats compile in parallel.
#include "ifa_decode_persist_launch.h"

namespace ifa {

bool dec_persist_has(int w_dtype, int nja, int njb, int hd)
{
    switch (w_dtype) {
    case Q4_B32T1A: case Q4_B32T1B: return dec_persist_has_dt<Q4_B32T1A>(nja, njb, hd);
    case Q3H_B64T1: return dec_persist_has_dt<Q3H_B64T1>(nja, njb, hd);
    default: return false;
    }
}

int dec_persist_launch(int w_dtype, int nja, int njb, int hd, int kvq8, const PsParams &P, int ncu, size_t smem, hipStream_t s)
{
    switch (w_dtype) {
    case Q4_B32T1A: case Q4_B32T1B: return dec_persist_launch_dt<Q4_B32T1A>(nja, njb, hd, kvq8, P, ncu, smem, s);
    case Q3H_B64T1: return dec_persist_launch_dt<Q3H_B64T1>(nja, njb, hd, kvq8, P, ncu, smem, s);
    default: return ifa_fail(IFA_ERR_DTYPE, "persistent decode: dtype %d", w_dtype);
    }
}

} // namespace ifa

extern "C" int ifa_experimental_built(void) { return 1; }
