// ifa_gemm_big.h -- the large-tile prefill GEMM (ifa_gemm.hip, k_gemm_big): T > 128 rows, weights dequantised once per
// workgroup and step into LDS, up to three matrices per launch, residual / GLU epilogues.
#pragma once
#include "ifa_gemm_rows_mfma.h"

namespace ifa {

// P: the argument block of the rows GEMM (GmArgs) with W[] / W1 pointing at REFERENCE-layout rows (Tensor::data);
// norm prologue fields are ignored.  epi: GM_PLAIN | GM_RESIDUAL | GM_GLU (the fused epilogues: Q4_B32T1A / B only).
bool gemm_big_ok(int w_dtype, const GmArgs &P, int epi);
int gemm_big(int w_dtype, const GmArgs &P, int epi, hipStream_t s);

} // namespace ifa
