// ifa_gemm_big.h -- the large-tile prefill GEMM (ifa_gemm.hip, k_gemm_big): T > 128 rows, weights dequantised once per
// workgroup and step into LDS, up to three matrices per launch, residual / GLU epilogues.
#pragma once
#include "ifa_gemm_rows_mfma.h"

namespace ifa {

// P: the argument block of the rows GEMM (GmArgs) with W[] / W1 pointing at REFERENCE-layout rows (Tensor::data);
// norm prologue fields are ignored.  epi: GM_PLAIN | GM_RESIDUAL | GM_GLU (the fused epilogues: Q4_B32T1A / B only).
struct BigGeo { int tile0[4]; int tiles_m; int K; int tn0; unsigned long long *part; unsigned *flags; unsigned *err;
                int sk_full, sk_rem, sk_tiles_n, sk_slots;
                int dbg_skip; };      // stream-K schedule (k_gemm_big<.., KS = 0>): whole tiles per workgroup, tiles shared by K steps, weight tiles of the launch, scratch slots per shared tile; dbg_skip (tests, bit 14 of ifa_gemm_big_tiles): the first parts of a split-K launch leave without publishing -- the last part's bounded wait must time out, leave its code and the host must fail the call and switch the waiting launches off
// per-(device, stream) scratch of the split-K launches: one counter per tile (zero between launches, 16 KB) in front of the partial sums
constexpr size_t SPLITK_FLAG_BYTES_H = 16384;
int gemm_splitk_scratch(hipStream_t s, size_t part_bytes, size_t tiles, void **out);

bool gemm_big_ok(int w_dtype, const GmArgs &P, int epi);
int gemm_big(int w_dtype, const GmArgs &P, int epi, hipStream_t s);

// Prompts of 33..512 tokens (ifa_gemm_mid.hip, k_gemm_mid): 128 x 128 tiles, weights in the MO layout (Tensor::mo, P.mo = 1),
// everything global -> LDS through a ring of direct-to-LDS stages, the weights dequantised from their raw bytes straight into
// the MFMA operand registers.  Same products, summation order over K and epilogues as gemm_big's 128 x 128 tiles.
bool gemm_mid_ok(int w_dtype, const GmArgs &P, int epi);
int gemm_mid(const GmArgs &P, int epi, hipStream_t s);

} // namespace ifa
