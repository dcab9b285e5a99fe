// chained decode launches (ifa_decode_chain.h) for Q4_B32T1A/B weights + the format dispatcher
#include <cstring>
#include "ifa_dchain_impl.h"

namespace ifa {

extern template int dec_chain_launch_dt<Q3H_B64T1>(bool, int, bool, const DecGemvParams &, const DecGemvParams &, const DecGemvParams &, const DecChainExtra &, int, hipStream_t);
template int dec_chain_launch_dt<Q4_B32T1A>(bool, int, bool, const DecGemvParams &, const DecGemvParams &, const DecGemvParams &, const DecChainExtra &, int, hipStream_t);

static bool ch_q4(int a) { return a == Q4_B32T1A || a == Q4_B32T1B; }
static bool ch_same(int a, int b) { return a == b || (ch_q4(a) && ch_q4(b)); }

// Which layers take it: W1, W3, W2 (and Wo) of one int8-path format with an instance, one workgroup per CU, one Wo row per wave,
// <= 2 W2 rows per loader wave; the instances cover dim 4096 with ffn 11008 / 14336 (Llama-2-7B, Mixtral's dense shape)
bool dec_chain_supported(int w_dtype, int w2_dtype, int wo_dtype, int dim, int ffn, bool wo, int wo_cols, int num_cus)
{
    if (!ch_q4(w_dtype) && w_dtype != Q3H_B64T1) return false;
    if (!ch_same(w_dtype, w2_dtype) || (wo && (!ch_same(w_dtype, wo_dtype) || wo_cols != dim))) return false;
    const int cap = block_capacity(w_dtype);
    if (dim % cap != 0 || ffn % cap != 0 || dim % 8 != 0 || ffn % 8 != 0) return false;
    const int nja = (dim / cap + 63) / 64, njb = (ffn / cap + 63) / 64;
    const bool inst = ch_q4(w_dtype) ? (nja == 2 && (njb == 6 || njb == 7)) : (nja == 1 && (njb == 3 || njb == 4));
    if (!inst) return false;
    const int W = num_cus * (CHAIN_TH / 64), WL = num_cus * (CHAIN_TH / 128);
    return num_cus >= 1 && num_cus <= 1024 && (dim + W - 1) / W == 1 && (dim + WL - 1) / WL <= 2;
}

int dec_chain_launch(int w_dtype, bool glu, int norm, bool wo, const DecGemvParams &P0, const DecGemvParams &Q0, const DecGemvParams *PW0,
                     const DecChainExtra &E, int num_cus, hipStream_t s)
{
    DecGemvParams P = P0, Q = Q0, PW;
    if (PW0) PW = *PW0; else memset(&PW, 0, sizeof(PW));
    const int cap = block_capacity(w_dtype);
    P.trace = nullptr; P.nsets = 1; P.total_rows = P.rows[0]; P.nblk = P.cols / cap;
    Q.trace = nullptr; Q.nsets = 1; Q.total_rows = Q.rows[0]; Q.nblk = Q.cols / cap;
    PW.trace = nullptr; PW.nsets = 1; PW.total_rows = PW.rows[0]; PW.nblk = PW.cols / cap;
    switch (w_dtype) {
    case Q4_B32T1A: case Q4_B32T1B: return dec_chain_launch_dt<Q4_B32T1A>(glu, norm, wo, P, Q, PW, E, num_cus, s);
    case Q3H_B64T1: return dec_chain_launch_dt<Q3H_B64T1>(glu, norm, wo, P, Q, PW, E, num_cus, s);
    default: return ifa_fail(IFA_ERR_DTYPE, "chained FFN launch: dtype %d", w_dtype);
    }
}

} // namespace ifa
