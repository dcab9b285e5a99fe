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
// bytes one block occupies in the tiled layout: the reference block's, except Q3H_B64T1 (see below)
__host__ __device__ constexpr int tiled_block_bytes(int dtype) { return dtype == Q3H_B64T1 ? 36 : block_bytes(dtype); }      // (Q3H_NATIVE: 32)

__host__ __device__ inline size_t tiled_row_bytes(int dtype, size_t nblk)
{
    return (nblk * (size_t)tiled_block_bytes(dtype) + 15) / 16 * 16;
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
// Q3H_B64T1 {base, scale, data_h[4], data_m[8], data[16]}: NOT a copy of planes -- the 32 seven-bit pair codes are
// expanded at load time into 32 nibble pairs (q3h_aos_to_nibbles below), which makes the streamed block
// [nibbles 32][base,scale] = the Q4_B64T1 tiled block, 36 bytes instead of 32.  (plane table = Q4_B64T1's; the
// source offsets are unused for this format)
IFA_TILED(Q3H_B64T1, 2, IFA_ARR(32, 4), IFA_ARR(4, 0))
// Q6_B64T1 {base, scale, data_h[16], data[32]}                   -> [data32][data_h16][base,scale]
IFA_TILED(Q6_B64T1, 3, IFA_ARR(32, 16, 4), IFA_ARR(20, 4, 0))
// Q5_B64T1 {base, scale, data_h[8], data[32]}                    -> [data32][data_h8][base,scale]
IFA_TILED(Q5_B64T1, 3, IFA_ARR(32, 8, 4), IFA_ARR(12, 4, 0))

#undef IFA_TILED
#undef IFA_ARR

// ---- Q3H_B64T1: the reference packs two 11-level codes into one 7-bit pair code p = q0 + 11 * q1 and splits p over
// three bit planes (4 + 2 + 1 bits, quantization.h:823-851).  Decoding that in the GEMV is VALU work per weight
// (re-assembling the planes: ~4 ops per element; with whole pair codes per byte and a multiply-shift division by 11:
// ~1.6) and kept the 3.5-bit format BEHIND the 4-bit one on a part where the stream should be the only cost (round 2:
// 607 tok/s vs 695 while reading 19 % fewer bytes).  The streaming layout is therefore chosen for decode cost: each
// pair code is expanded ONCE, at load time, into the byte q0 | q1 << 4 -- exactly the Q4_B64T1 convention (byte i =
// elements 2i low, 2i + 1 high) with codes 0..10 -- so the fused kernels decode it with two nibble masks and the
// block is 36 bytes per 64 weights (4.5 bits) instead of 32 (4.0): still 10 % fewer streamed bytes than Q4_B32T1
// (40 per 64).  bench.py counts the 36.  The interchange format (ifa_quantize / ifa_dequantize, goldens, prefill)
// stays the reference's 32-byte block; q3h_nibbles_to_aos is the inverse (lossless).
__host__ __device__ inline void q3h_aos_to_nibbles(const uint8_t *aos, uint8_t *n32)
{
    uint8_t p[32];
    // (q3h_pairs_from_aos is defined below)
    for (int idx = 0; idx < 8; idx++) {
        const uint32_t u16v = (uint32_t)aos[16 + 2 * idx] | ((uint32_t)aos[17 + 2 * idx] << 8);
        const uint32_t m8 = aos[8 + idx];
        const uint32_t hb = aos[4 + idx / 2];
        const uint32_t h8 = (idx % 2 == 0) ? (hb & 0x0F) : (hb >> 4);
        for (int n = 0; n < 4; n++)
            p[4 * idx + n] = (uint8_t)(((u16v >> (4 * n)) & 0xF) | (((m8 >> (2 * n)) & 3) << 4) | (((h8 >> n) & 1) << 6));
    }
    for (int k = 0; k < 32; k++) n32[k] = (uint8_t)((p[k] % 11) | ((p[k] / 11) << 4));
}

// Older byte-transposed form (round 1-2), kept for q3h_tiled_to_aos users:
//   D[w] byte b (w = 0..6) = p[4w+b] | bit w of p[28+b] << 7        (p[k] = code of elements 2k, 2k+1)
__host__ __device__ inline void q3h_pairs_from_aos(const uint8_t *aos, uint8_t *p)
{
    for (int idx = 0; idx < 8; idx++) {
        const uint32_t u16v = (uint32_t)aos[16 + 2 * idx] | ((uint32_t)aos[17 + 2 * idx] << 8);
        const uint32_t m8 = aos[8 + idx];
        const uint32_t hb = aos[4 + idx / 2];
        const uint32_t h8 = (idx % 2 == 0) ? (hb & 0x0F) : (hb >> 4);
        for (int n = 0; n < 4; n++)
            p[4 * idx + n] = (uint8_t)(((u16v >> (4 * n)) & 0xF) | (((m8 >> (2 * n)) & 3) << 4) | (((h8 >> n) & 1) << 6));
    }
}

__host__ __device__ inline void q3h_aos_to_tiled(const uint8_t *aos, uint8_t *d28)   // d28: D0..D6 as 28 bytes
{
    uint8_t p[32];
    q3h_pairs_from_aos(aos, p);
    for (int w = 0; w < 7; w++)
        for (int b = 0; b < 4; b++) d28[4 * w + b] = (uint8_t)(p[4 * w + b] | (((p[28 + b] >> w) & 1) << 7));
}

__host__ __device__ inline void q3h_tiled_to_aos(const uint8_t *d28, uint8_t *aos)   // fills aos[4..31]
{
    uint8_t p[32];
    for (int b = 0; b < 4; b++) p[28 + b] = 0;
    for (int w = 0; w < 7; w++)
        for (int b = 0; b < 4; b++) {
            p[4 * w + b] = d28[4 * w + b] & 0x7F;
            p[28 + b] = (uint8_t)(p[28 + b] | ((d28[4 * w + b] >> 7) << w));
        }
    for (int i = 4; i < 32; i++) aos[i] = 0;
    for (int idx = 0; idx < 8; idx++)
        for (int n = 0; n < 4; n++) {
            const uint32_t v = p[4 * idx + n];
            const uint32_t nib = (v & 0xF) << (4 * n);
            aos[16 + 2 * idx] = (uint8_t)(aos[16 + 2 * idx] | (nib & 0xFF));
            aos[17 + 2 * idx] = (uint8_t)(aos[17 + 2 * idx] | (nib >> 8));
            aos[8 + idx] = (uint8_t)(aos[8 + idx] | (((v >> 4) & 3) << (2 * n)));
            aos[4 + idx / 2] = (uint8_t)(aos[4 + idx / 2] | (((v >> 6) & 1) << (n + 4 * (idx % 2))));
        }
}

} // namespace ifa
