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
};

// rows of 16 per set when nsets > 1, cols % 128 == 0, 2 <= T <= 8; norm == 1 needs cols <= 4096
bool gemm_rows_mfma_fused_ok(const GmArgs &P, int epi, int norm);
int gemm_rows_mfma_launch(const GmArgs &P, int epi, int norm, hipStream_t s);

} // namespace ifa
