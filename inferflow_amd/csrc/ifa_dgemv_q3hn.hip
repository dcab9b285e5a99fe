// fused decode GEMV kernels for Q3H_B64T1 values at their native 32 bytes per block (Q3H_NATIVE: ifa_decode_formats.h WRowQ3HN;
// engine option q3h_native) + the load-time copy they stream
#include "ifa_decode_gemv_impl.h"
namespace ifa {

template int dec_gemv_launch_dt<Q3H_NATIVE>(int, int, const DecGemvParams &, int, hipStream_t);

// reference-layout rows (32-byte blocks {base, scale, data_h[4], data_m[8], data[16]}) -> [D0..D3][D4, D5][D6, (base, scale)] planes
__global__ void __launch_bounds__(256) k_q3h_native_rows(const uint8_t *__restrict__ aos, int rows, int nblk, uint8_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * nblk) return;
    const int row = (int)(i / nblk), blk = (int)(i % nblk);
    uint8_t b[32], d28[28];
    const uint32_t *s = reinterpret_cast<const uint32_t *>(aos + i * 32);
#pragma unroll
    for (int w = 0; w < 8; w++) { const uint32_t v = s[w]; b[4 * w] = (uint8_t)v; b[4 * w + 1] = (uint8_t)(v >> 8); b[4 * w + 2] = (uint8_t)(v >> 16); b[4 * w + 3] = (uint8_t)(v >> 24); }
    q3h_aos_to_tiled(b, d28);
    uint8_t *r = dst + (size_t)row * nblk * 32;
    uint32_t D[7];
#pragma unroll
    for (int w = 0; w < 7; w++) D[w] = (uint32_t)d28[4 * w] | ((uint32_t)d28[4 * w + 1] << 8) | ((uint32_t)d28[4 * w + 2] << 16) | ((uint32_t)d28[4 * w + 3] << 24);
    uint32_t *p0 = reinterpret_cast<uint32_t *>(r + (size_t)blk * 16);
    uint32_t *p1 = reinterpret_cast<uint32_t *>(r + (size_t)nblk * 16 + (size_t)blk * 8);
    uint32_t *p2 = reinterpret_cast<uint32_t *>(r + (size_t)nblk * 24 + (size_t)blk * 8);
    p0[0] = D[0]; p0[1] = D[1]; p0[2] = D[2]; p0[3] = D[3];
    p1[0] = D[4]; p1[1] = D[5];
    p2[0] = D[6]; p2[1] = s[0];
}

int q3h_native_rows(const void *aos, size_t rows, size_t cols, void *dst, hipStream_t s)
{
    IFA_REQUIRE(cols % 64 == 0 && rows > 0, "Q3H native copy: %zu x %zu", rows, cols);
    const size_t n = rows * (cols / 64);
    k_q3h_native_rows<<<ifa_cdiv(n, 256), 256, 0, s>>>((const uint8_t *)aos, (int)rows, (int)(cols / 64), (uint8_t *)dst);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

} // namespace ifa
