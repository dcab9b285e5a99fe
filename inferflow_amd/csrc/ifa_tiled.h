// ifa_tiled.h -- the "row-local plane" weight layout streamed by the fused
// decode kernels (DESIGN.md "HBM layout").
//
// A reference row is nblk AoS blocks {base, scale, bit-planes...}
// (src/common/quant_types.h).  Block sizes of 20/34/36/44/52 B defeat aligned
// 16-byte lane loads, so at load time each ROW is re-tiled (same byte count,
// same row stride) into planes:
//     row = [plane0 of block 0..nblk-1][plane1 of block 0..nblk-1]...
// plane0 is always the big low-bits plane (16 or 32 B per block) so that lane l
// reading block l+64j issues an aligned global_load_dwordx4 and a whole wave
// covers a contiguous 1-2 KiB.  The row stride is padded to 16 bytes (tiled_row_bytes).  The reference AoS layout stays the interchange
// format (ifa_quantize / ifa_dequantize / goldens); ifa_repack_weights converts.
#pragma once
#include "ifa_device.h"

namespace ifa {

template <int DT> struct TiledLayout;

// Row stride of the tiled layout: the reference row (nblk * block bytes) padded to a
// multiple of 16 so that every row's big plane starts 16-byte aligned for any width
// (e.g. 43 blocks per row when Llama-2-7B's W2 is column-sliced 8 ways).
__host__ __device__ inline size_t tiled_row_bytes(int dtype, size_t nblk)
{
    return (nblk * (size_t)block_bytes(dtype) + 15) / 16 * 16;
}

// each plane p copies AoS bytes [src_off, src_off+len)
#define IFA_TILED(DTV, N, LENS, OFFS)                                                        \
    template <> struct TiledLayout<DTV> {                                                     \
        static constexpr int NPLANES = N;                                                     \
        __host__ __device__ static constexpr int plane_len(int p) { constexpr int a[] = LENS; return a[p]; }     \
        __host__ __device__ static constexpr int plane_src_off(int p) { constexpr int a[] = OFFS; return a[p]; } \
        __host__ __device__ static constexpr int plane_start(int p)                           \
        { int s = 0; for (int i = 0; i < p; i++) s += plane_len(i); return s; }               \
    };
#define IFA_ARR(...) {__VA_ARGS__}

// Q4_B32T1 {base u16, scale u16, data[16]}                       -> [data16][base,scale]
IFA_TILED(Q4_B32T1A, 2, IFA_ARR(16, 4), IFA_ARR(4, 0))
IFA_TILED(Q4_B32T1B, 2, IFA_ARR(16, 4), IFA_ARR(4, 0))
// Q8_B32T2 {scale f16, data[32]}                                 -> [data32][scale]
IFA_TILED(Q8_B32T2, 2, IFA_ARR(32, 2), IFA_ARR(2, 0))
// Q4_B64T1 {base, scale, data[32]}                               -> [data32][base,scale]
IFA_TILED(Q4_B64T1, 2, IFA_ARR(32, 4), IFA_ARR(4, 0))
// Q3H_B64T1 {base, scale, data_h[4], data_m[8], data[16]}        -> [data16][data_m8][base,scale,data_h4]
IFA_TILED(Q3H_B64T1, 3, IFA_ARR(16, 8, 8), IFA_ARR(16, 8, 0))
// Q6_B64T1 {base, scale, data_h[16], data[32]}                   -> [data32][data_h16][base,scale]
IFA_TILED(Q6_B64T1, 3, IFA_ARR(32, 16, 4), IFA_ARR(20, 4, 0))
// Q5_B64T1 {base, scale, data_h[8], data[32]}                    -> [data32][data_h8][base,scale]
IFA_TILED(Q5_B64T1, 3, IFA_ARR(32, 8, 4), IFA_ARR(12, 4, 0))

#undef IFA_TILED
#undef IFA_ARR

} // namespace ifa
