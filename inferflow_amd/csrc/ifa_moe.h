// ifa_moe.h -- device-side work lists of a mixture-of-experts step over T > 1 rows (ProcessGpuLayer_Moe,
// src/transformer/inference_worker.cc:1924-2146; HostTensorOpr::BuildRowsForMoE, src/tensor/host_tensor_opr.cc:190-244).
// The reference copies the router probabilities to the host, builds per-expert row lists there and runs the experts one
// after the other; here the lists live on the device and ALL experts run in grouped launches:
//   entries   : (token row, weight) pairs in the reference's order -- experts ascending, token order inside an expert;
//               an expert's rows are contiguous in the gathered activation / output buffers
//   tiles     : <= 64 (or 128, for long prompts) consecutive entries of ONE expert that has >= 2 rows -> one MFMA GEMM tile (the reference's T > 1
//               branch of MatrixMultiplication: F16 activations on dequantised weights)
//   smalls    : experts with 2 .. small_max rows (dynamic batching: a handful of rows per expert) -> the same T > 1
//               arithmetic from the weight-streaming kernel (ifa_gemm_rows.hip) instead of a mostly empty MFMA tile
//   singles   : experts with exactly one row -> the int8-activation GEMV arithmetic (its T = 1 branch)
#pragma once
#include <stdint.h>

namespace ifa {

struct MoeTile { int expert, row0, nrows, pad; };
struct MoeSingle { int expert, pos; };

// counters written by k_moe_build: [0] entries, [1] tiles, [2] singles, [3] smalls
struct MoeGroup {
    const MoeTile *tiles;
    const MoeSingle *singles;
    const int *counts;
    const uint8_t *const *wtab;     // [expert][3] reference-layout (AoS) weight pointers {w1, w2, w3}
    int which;                      // 0 w1, 1 w2, 2 w3
    int on;                         // 0: plain (ungrouped) launch
};
// the small-group launch: its list and the TILED weight pointers ([expert][4]: {w1, w3, w2, -}, the fused decode kernels' table)
struct MoeSmallGroup {
    const MoeTile *smalls;
    const int *counts;
    const uint8_t *const *wtab_tiled;
    int which_tiled;                // 0 w1, 1 w3, 2 w2
    const uint8_t *const *wtab_mo;  // the same table of MO copies (ifa_gemm_rows_mfma.h), or null
};

} // namespace ifa
