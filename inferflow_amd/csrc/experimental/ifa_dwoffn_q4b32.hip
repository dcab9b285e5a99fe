// Wo rows in front of the W1 / W3 launch for Q4_B32T1A/B weights + the format dispatcher (see ifa_decode_wo_ffn.h)
#include "ifa_dwoffn_impl.h"

namespace ifa {

extern template int dec_wo_ffn_launch_dt<Q3H_B64T1>(bool, const DecGemvParams &, const DecGemvParams &, const DecWoFfnExtra &, int, hipStream_t);
template int dec_wo_ffn_launch_dt<Q4_B32T1A>(bool, const DecGemvParams &, const DecGemvParams &, const DecWoFfnExtra &, int, hipStream_t);

static bool wf_same(int a, int b) { const bool qa = a == Q4_B32T1A || a == Q4_B32T1B, qb = b == Q4_B32T1A || b == Q4_B32T1B; return a == b || (qa && qb); }

// Which layers take it: Wo, W1 (and W3) of one int8-path format with an instance; 4096-column rows on both sides (dim ==
// heads * head_dim == 4096); one workgroup per CU, the FFN rows dealt <= 3 pairs and the Wo rows <= 2 per wave.
bool dec_wo_ffn_supported(int w_dtype, int wo_dtype, int w3_dtype, int dim, int wo_cols, int ffn_rows, bool glu, int num_cus)
{
    const bool q4 = w_dtype == Q4_B32T1A || w_dtype == Q4_B32T1B;
    if (!q4 && w_dtype != Q3H_B64T1) return false;
    if (!wf_same(w_dtype, wo_dtype) || (glu && !wf_same(w_dtype, w3_dtype))) return false;
    if (dim != 4096 || wo_cols != 4096) return false;
    if (num_cus < 2 * WF_FRONT || num_cus % WF_FRONT != 0) return false;               // front groups of 8 workgroups, evenly spread
    if (dim != WF_FRONT * (WF_THREADS / 64) * 4) return false;                        // 4 Wo rows per front wave
    const int wl = (num_cus - WF_FRONT) * (WF_THREADS / 64);                          // loader waves of the grid
    if (ffn_rows < 1 || ffn_rows > 3 * wl + 2 * WF_FRONT * (WF_THREADS / 64)) return false;     // 3 row pairs per loader wave, <= 2 per front wave
    return true;
}

int dec_wo_ffn_launch(int w_dtype, bool glu, const DecGemvParams &PW0, const DecGemvParams &P0, const DecWoFfnExtra &E, int num_cus, hipStream_t s)
{
    DecGemvParams PW = PW0, P = P0;
    PW.trace = nullptr; PW.nsets = 1; PW.total_rows = PW.rows[0]; PW.nblk = PW.cols / block_capacity(w_dtype);
    P.trace = nullptr; P.nsets = 1; P.total_rows = P.rows[0]; P.nblk = P.cols / block_capacity(w_dtype);
    switch (w_dtype) {
    case Q4_B32T1A: case Q4_B32T1B: return dec_wo_ffn_launch_dt<Q4_B32T1A>(glu, PW, P, E, num_cus, s);
    case Q3H_B64T1: return dec_wo_ffn_launch_dt<Q3H_B64T1>(glu, PW, P, E, num_cus, s);
    default: return ifa_fail(IFA_ERR_DTYPE, "fused Wo + FFN launch: dtype %d", w_dtype);
    }
}

} // namespace ifa
