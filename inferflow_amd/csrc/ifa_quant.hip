// ifa_quant.hip -- TensorOpr::Quantize / Dequantize counterparts
// (src/tensor/tensor_opr.cu:1623-2305, kernels src/kernels/tensor_quant.h) and
// the load-time re-tiling of weight rows for the fused decode kernels.
#include "ifa_host.h"
#include "ifa_codec.h"
#include "ifa_tiled.h"

namespace ifa {

// one thread per block (load time; reference: tensor_quant.h:8-299 also one thread per block)
template <int DT, typename SrcT>
__global__ void __launch_bounds__(256) k_quantize_blocks(const SrcT *__restrict__ src, uint8_t *__restrict__ dst,
                                                         size_t total_blocks)
{
    constexpr int CAP = block_capacity(DT), BB = block_bytes(DT);
    size_t bi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (bi >= total_blocks) return;
    float s[CAP];
    const SrcT *p = src + bi * CAP;
#pragma unroll
    for (int i = 0; i < CAP; i++) s[i] = (float)p[i];
    RawBlock<BB> b;
    quantize_block<DT>(s, b);
    b.store(dst + bi * BB);
}

// 256 blocks per workgroup: the packed bytes arrive through LDS with coalesced loads (a thread-per-block walk of
// 20..68-byte structs touches every line BB/2 times), one thread decodes one block, values leave as 16-byte stores
template <int DT>
__global__ void __launch_bounds__(256) k_dequantize_blocks(const uint8_t *__restrict__ src, half_t *__restrict__ dst,
                                                           size_t total_blocks)
{
    constexpr int CAP = block_capacity(DT), BB = block_bytes(DT);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) uint16_t raw[256 * BB / 2];
    const int tid = threadIdx.x;
    const size_t b0 = (size_t)blockIdx.x * 256;
    const int n = (int)min((size_t)256, total_blocks - b0);
    const uint8_t *s = src + b0 * BB;
    const int nh = n * BB / 2;                              // 16-bit words of this workgroup's blocks
    if ((reinterpret_cast<uintptr_t>(s) & 3) == 0) {
        for (int i = tid; i < nh / 2; i += 256) reinterpret_cast<uint32_t *>(raw)[i] = reinterpret_cast<const uint32_t *>(s)[i];
        if ((nh & 1) && tid == 0) raw[nh - 1] = reinterpret_cast<const uint16_t *>(s)[nh - 1];
    } else {
        for (int i = tid; i < nh; i += 256) raw[i] = reinterpret_cast<const uint16_t *>(s)[i];
    }
    __syncthreads();
    if (tid >= n) return;
    RawBlock<BB> b;
    b.load(reinterpret_cast<const uint8_t *>(raw) + tid * BB);
    int q[CAP]; float scale, base;
    decode_block<DT>(b, q, scale, base);
    half_t *o = dst + (b0 + tid) * CAP;
    if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) {
#pragma unroll
        for (int c = 0; c < CAP / 8; c++) {
            half_t v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = f2h(block_value<DT>(q[8 * c + e], scale, base));
            *reinterpret_cast<u32x4 *>(o + 8 * c) = *reinterpret_cast<const u32x4 *>(v);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CAP; i++) o[i] = f2h(block_value<DT>(q[i], scale, base));
    }
}

// Tensor_QuantizeQ8_B32T2_Alg2_Kernel (src/kernels/tensor_quant.h:44-82): one
// lane per element; the 32-lane CUDA warp max becomes a half-wave max.
__global__ void __launch_bounds__(256) k_quantize_act_q8(const half_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                         int cols, int dst_row_bytes)
{
    const int row = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const float value = col < cols ? h2f(src[(size_t)row * cols + col]) : 0.0f;
    const float mx = half_wave_max(fabsf(value));
    const float scale = mx / 127;
    int q = scale <= 0.000001f ? 0 : (int)roundf(value / scale);
    q = min(max(q, -128), 127);
    if (col < cols) {
        uint8_t *blk = dst + (size_t)row * dst_row_bytes + (size_t)(col / 32) * 34;
        blk[2 + (col & 31)] = (uint8_t)(int8_t)q;
        if ((col & 31) == 0) *reinterpret_cast<uint16_t *>(blk) = f2hbits(scale);
    }
}

// AoS -> row-local planes (ifa_tiled.h), one thread per block
template <int DT>
__global__ void __launch_bounds__(256) k_repack(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                int rows, int nblk)
{
    using L = TiledLayout<DT>;
    constexpr int BB = block_bytes(DT);
    size_t bi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (bi >= (size_t)rows * nblk) return;
    int row = (int)(bi / nblk), blk = (int)(bi % nblk);
    const uint8_t *s = src + bi * BB;
    uint8_t *drow = dst + (size_t)row * tiled_row_bytes(DT, (size_t)nblk);
    if constexpr (DT == Q3H_B64T1) {      // pair codes expanded to nibble pairs: the Q4_B64T1 tiled block (ifa_tiled.h)
        uint8_t a[32], n32[32];
        for (int i = 0; i < 32; i++) a[i] = s[i];
        q3h_aos_to_nibbles(a, n32);
        uint8_t *p0 = drow + (size_t)blk * 32, *p1 = drow + (size_t)32 * nblk + (size_t)blk * 4;
        for (int i = 0; i < 32; i++) p0[i] = n32[i];
        for (int i = 0; i < 4; i++) p1[i] = a[i];
        return;
    }
#pragma unroll
    for (int p = 0; p < L::NPLANES; p++) {
        uint8_t *d = drow + (size_t)L::plane_start(p) * nblk + (size_t)blk * L::plane_len(p);
        for (int i = 0; i < L::plane_len(p); i += 2)
            *reinterpret_cast<uint16_t *>(d + i) = *reinterpret_cast<const uint16_t *>(s + L::plane_src_off(p) + i);
    }
}

} // namespace ifa

using namespace ifa;

template <typename SrcT>
static int quantize_impl(int dtype, const SrcT *src, size_t rows, size_t cols, void *dst, ifa_stream stream)
{
    IFA_REQUIRE(src && dst, "ifa_quantize: null pointer");
    int cap = block_capacity(dtype);
    IFA_REQUIRE(cap > 1, "ifa_quantize: dtype %d is not a block format", dtype);
    IFA_REQUIRE(cols % (size_t)cap == 0, "ifa_quantize: cols %zu not a multiple of block capacity %d", cols, cap);
    size_t total = rows * (cols / (size_t)cap);
    if (total == 0) return IFA_OK;
    IFA_DISPATCH_QUANT_DTYPE(dtype, k_quantize_blocks<DT, SrcT><<<dim3(ifa_cdiv(total, 256)), dim3(256), 0, ifa_s(stream)>>>(src, (uint8_t *)dst, total));
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

extern "C" {

int ifa_quantize(int dtype, const void *src_f16, size_t rows, size_t cols, void *dst, ifa_stream stream)
{
    return quantize_impl<half_t>(dtype, (const half_t *)src_f16, rows, cols, dst, stream);
}

int ifa_quantize_f32(int dtype, const void *src_f32, size_t rows, size_t cols, void *dst, ifa_stream stream)
{
    return quantize_impl<float>(dtype, (const float *)src_f32, rows, cols, dst, stream);
}

int ifa_dequantize(int dtype, const void *src, size_t rows, size_t cols, void *dst_f16, ifa_stream stream)
{
    IFA_REQUIRE(src && dst_f16, "ifa_dequantize: null pointer");
    int cap = block_capacity(dtype);
    IFA_REQUIRE(cap > 1, "ifa_dequantize: dtype %d is not a block format", dtype);
    IFA_REQUIRE(cols % (size_t)cap == 0, "ifa_dequantize: cols %zu not a multiple of block capacity %d", cols, cap);
    size_t total = rows * (cols / (size_t)cap);
    if (total == 0) return IFA_OK;
    IFA_DISPATCH_QUANT_DTYPE(dtype, k_dequantize_blocks<DT><<<dim3(ifa_cdiv(total, 256)), dim3(256), 0, ifa_s(stream)>>>((const uint8_t *)src, (half_t *)dst_f16, total));
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_quantize_act_q8(const void *src_f16, size_t rows, size_t cols, void *dst, ifa_stream stream)
{
    IFA_REQUIRE(src_f16 && dst, "ifa_quantize_act_q8: null pointer");
    if (rows == 0 || cols == 0) return IFA_OK;
    IFA_REQUIRE(rows <= 65535, "ifa_quantize_act_q8: too many rows (%zu)", rows);
    int row_bytes = (int)((cols + 31) / 32 * 34);
    k_quantize_act_q8<<<dim3(ifa_cdiv(cols, 256), (unsigned)rows), dim3(256), 0, ifa_s(stream)>>>((const half_t *)src_f16, (uint8_t *)dst, (int)cols, row_bytes);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

size_t ifa_tiled_row_bytes(int dtype, size_t cols)
{
    int cap = block_capacity(dtype);
    if (cap <= 1 || !ax8_eligible(dtype)) return 0;
    return tiled_row_bytes(dtype, (cols + (size_t)cap - 1) / (size_t)cap);
}

int ifa_repack_weights(int dtype, const void *src, size_t rows, size_t cols, void *dst, ifa_stream stream)
{
    IFA_REQUIRE(src && dst && src != dst, "ifa_repack_weights: null or aliased pointers");
    IFA_REQUIRE(ax8_eligible(dtype), "ifa_repack_weights: dtype %d has no tiled layout", dtype);
    int cap = block_capacity(dtype);
    IFA_REQUIRE(cols % (size_t)cap == 0, "ifa_repack_weights: cols %zu not a multiple of %d", cols, cap);
    size_t nblk = cols / (size_t)cap, total = rows * nblk;
    if (total == 0) return IFA_OK;
#define IFA_REPACK_CASE(T) case T: k_repack<T><<<dim3(ifa_cdiv(total, 256)), dim3(256), 0, ifa_s(stream)>>>(\
                                                      (const uint8_t *)src, (uint8_t *)dst, (int)rows, (int)nblk); break;
    switch (dtype) {
        IFA_REPACK_CASE(Q8_B32T2) IFA_REPACK_CASE(Q6_B64T1) IFA_REPACK_CASE(Q5_B64T1) IFA_REPACK_CASE(Q4_B32T1A)
        IFA_REPACK_CASE(Q4_B32T1B) IFA_REPACK_CASE(Q4_B64T1) IFA_REPACK_CASE(Q3H_B64T1)
    default: return ifa_fail(IFA_ERR_DTYPE, "ifa_repack_weights: dtype %d", dtype);
    }
#undef IFA_REPACK_CASE
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

} // extern "C"
