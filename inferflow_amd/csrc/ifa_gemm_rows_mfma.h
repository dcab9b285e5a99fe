// ifa_gemm_rows_mfma.h -- launch interface of the 2..8-row weight-streaming GEMM on the matrix cores
// (ifa_gemm_rows_mfma.hip), with the prologue / epilogues of the fused batched decode step.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "ifa_device.h"

namespace ifa {

// Arguments.  Up to three matrices ("sets": wq | wk | wv) form one virtual row space (each set's rows % 16 == 0 when there
// is more than one); GM_GLU pairs W[0] (w1) with W1 (w3).  Y / res are row-major over the VIRTUAL rows with their own strides.
enum GmEpilogue { GM_PLAIN = 0, GM_RESIDUAL = 1, GM_GLU = 2 };
struct GmArgs {
    const uint8_t *W[3];
    const uint8_t *W1;
    int rows[3];
    int nsets, total_rows, nblk, T;
    const half_t *X;
    int ldx;
    float multi_base, eps;
    const half_t *norm_w;          // NORM == 1: RMS weight ([K], may be null); the rows are normalised while they are staged
    const half_t *bias[3];
    const half_t *bias1;
    half_t *Y;
    const half_t *res;             // GM_RESIDUAL: Y = half(res + y)  (TensorOpr::Add)
    int ldy, ldres;
    int act_kind;
    half_t *Yset[3];               // optional: set i written to its own matrix Yset[i][t * ldyset[i] + row] (q, k, v of a prompt)
    int ldyset[3];
    int mo;                        // W[] / W1 are MO-layout copies (below) instead of the tiled layout
    long long *trace;              // optional [workgroups][16] wall-clock stamps (tuning: IFA_ROWS_TRACE=1 prints a timeline per launch)
    // K parts (round 4: chunked rows of 9..32 queries, kernels instantiated with KPM > 0): kparts consecutive workgroups share a
    // group of KPM * kparts tiles; each walks its own chunks of K for ALL of them, leaves the fp32 sums of the tiles it does not
    // finish in kpart_sums as 8-byte {1, value} granules (zero between launches) and finishes tiles [kp * KPM, (kp + 1) * KPM):
    // its own part from LDS, the others' granules polled, added in K order.  Filled by the launcher; 0 = off.
    int kparts;
    unsigned long long *kpart_sums;
    // in-launch waits (ifa_host.h): the device's wait-error word (filled by the launchers that wait: K parts here, the split-K
    // halves of k_gemm_big), and the caller's veto -- no_waits != 0: never pick a launch that waits (model options rows_kparts /
    // gemm_splitk = 0)
    unsigned *wait_err;
    int no_waits;
};

// MO layout ("MFMA operand order") of a matrix [rows][cols] of 4-bit codes with value q * scale + base, cols % 128 == 0: per tile of 16 rows
//   nsup = cols / 128 supersteps of 1024 bytes: lane l = 16 g + r of a wave owns bytes 16 l .. 16 l + 15 = the 16 code bytes of
//   block 4 S + g of row 16 tile + r (exactly its A operands of the superstep's four MFMAs), then
//   ceil(nsup / 4) quads of 1024 bytes: lane l owns the (base, scale) words of its blocks in supersteps 4 Q .. 4 Q + 3.
// Rows past the end and the pad of the last quad are zero.  Same size as the tiled copy (+ the pad).
size_t gemm_rows_mo_bytes(size_t rows, size_t cols);
// dtype: Q4_B32T1A / B, or the 64-weight nibble formats Q4_B64T1 / Q3H_B64T1 (one (base, scale) per 64 weights, written for both
// 32-weight halves: the kernel then runs unchanged; cols % 128 == 0)
int gemm_rows_mo_build(int dtype, const void *tiled, size_t rows, size_t cols, void *mo, hipStream_t s);

// rows of 16 per set when nsets > 1, cols % 128 == 0, 2 <= T <= 8; norm == 1 needs cols <= 4096
// the per-stream scratch of the K-parts launches exists (call before capturing a step that may use them)
int gemm_rows_kparts_reserve(hipStream_t s);
bool gemm_rows_mfma_fused_ok(const GmArgs &P, int epi, int norm);
int gemm_rows_mfma_launch(const GmArgs &P, int epi, int norm, hipStream_t s);

// Q4_B64T1 / Q3H_B64T1 (tiled, as streamed) -> Q4_B32T1A reference-layout rows of cols / 32 twenty-byte blocks (same values)
int expand_b64_to_q4b32(int dtype, const void *tiled, size_t rows, size_t cols, void *out_aos, hipStream_t s);

} // namespace ifa
