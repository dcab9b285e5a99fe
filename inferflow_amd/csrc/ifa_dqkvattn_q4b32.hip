// fused QKV + attention launch for Q4_B32T1A/B weights + the format dispatcher (see ifa_decode_qkv_attn.h)
#include <algorithm>
#include "ifa_dqkvattn_impl.h"

namespace ifa {

extern template int dec_qkv_attn_launch_dt<Q3H_B64T1>(int, bool, int, bool, int, const DecGemvParams &, const DecAttnParams &, const DecQkvAttnExtra &, int, hipStream_t);
template int dec_qkv_attn_launch_dt<Q4_B32T1A>(int, bool, int, bool, int, const DecGemvParams &, const DecAttnParams &, const DecQkvAttnExtra &, int, hipStream_t);

// Which models take the fused launch: q / k / v of one int8-path format with a kernel instance; 128-wide heads; a grid of one
// workgroup per CU that splits evenly into kv groups (gk workgroups each, a multiple of the query heads per group) whose
// rows deal out evenly (rw per wave).
bool dec_qkv_attn_supported(int w_dtype, int cols, int heads, int kv_heads, int head_dim, int num_cus, int *gk_out, int *rw_out)
{
    const bool q4 = w_dtype == Q4_B32T1A || w_dtype == Q4_B32T1B;
    if (!q4 && w_dtype != Q3H_B64T1) return false;
    if (head_dim != 128 || cols != 4096 || kv_heads < 1 || heads % kv_heads != 0) return false;
    if (num_cus < kv_heads || num_cus % kv_heads != 0) return false;
    const int gk = num_cus / kv_heads, group = heads / kv_heads;
    if (gk % group != 0) return false;
    const int rg = (group + 2) * head_dim, wv = gk * (QA_THREADS / 64);
    if (rg % wv != 0) return false;
    const int rw = rg / wv;
    if (rw != 6 && rw != 3) return false;
    if (gk_out) *gk_out = gk;
    if (rw_out) *rw_out = rw;
    return true;
}

int dec_qkv_attn_launch(int w_dtype, int norm, bool kv_q8, int pb, bool kt, const DecGemvParams &P0, const DecAttnParams &A, const DecQkvAttnExtra &E,
                        int max_ctx, hipStream_t s)
{
    DecGemvParams P = P0;
    P.trace = nullptr;
    P.total_rows = P.rows[0] + P.rows[1] + P.rows[2];
    P.nblk = P.cols / block_capacity(w_dtype);
    int gk = 0, rw = 0;
    if (!dec_qkv_attn_supported(w_dtype, P.cols, A.heads, A.kv_heads, 128, A.kv_heads * E.gk, &gk, &rw) || gk != E.gk)
        return ifa_fail(IFA_ERR_ARG, "fused QKV + attention: unsupported shape");
    switch (w_dtype) {
    case Q4_B32T1A: case Q4_B32T1B: return dec_qkv_attn_launch_dt<Q4_B32T1A>(norm, kv_q8, pb, kt, rw, P, A, E, max_ctx, s);
    case Q3H_B64T1: return dec_qkv_attn_launch_dt<Q3H_B64T1>(norm, kv_q8, pb, kt, rw, P, A, E, max_ctx, s);
    default: return ifa_fail(IFA_ERR_DTYPE, "fused QKV + attention: dtype %d", w_dtype);
    }
}

} // namespace ifa
