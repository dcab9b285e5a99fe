// ref_quant_wrap.cc -- thin extern "C" shim around the REFERENCE's own block
// codecs, compiled from the sources where they lie (never copied):
//   $(REF)/src/common/quantization.h, quant_types.h, data_types.h,
//   $(REF)/3rd_party/half/half.hpp
// Output: oracle/_ref/libifa_ref_quant.so (git-ignored; travels to the GPU box).
// Test infrastructure only: pins oracle/ifa_oracle.c bit-for-bit and generates
// tests/golden/*.npz (tests/golden/gen_golden.py).  Nothing in the product
// path may load it.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include "common/quantization.h"

using namespace inferflow;

namespace {
// numeric ids = reference ElementType enum values (src/tensor/tensor_common.h:15-42)
enum {
    T_Q8_B32T1 = 7, T_Q8_B32T2 = 8, T_Q6_B64T1 = 9, T_Q5_B64T1 = 10, T_Q5_B32T1 = 11,
    T_Q4_B16 = 12, T_Q4_B32T1A = 13, T_Q4_B32T1B = 14, T_Q4_B64T1 = 17, T_Q3H_B64T1 = 18,
    T_Q3_B32T1A = 19, T_Q3_B32T1B = 20, T_Q2_B32T1A = 21, T_Q2_B32T1B = 22
};

template <typename S>
int quantize_row(int dtype, const S *src, int cols, uint8_t *dst)
{
    switch (dtype) {
    case T_Q8_B32T1: return Quantization::QuantizeRow_Q8_B32T1((BlockQ8_B32T1*)dst, cols / 32, src, cols) ? 0 : -1;
    case T_Q8_B32T2: return Quantization::QuantizeRow_Q8_B32T2((BlockQ8_B32T2*)dst, cols / 32, src, cols) ? 0 : -1;
    case T_Q6_B64T1: return Quantization::QuantizeRow_Q6_B64T1((BlockQ6_B64T1*)dst, cols / 64, src, cols) ? 0 : -1;
    case T_Q5_B64T1: return Quantization::QuantizeRow_Q5_B64T1((BlockQ5_B64T1*)dst, cols / 64, src, cols) ? 0 : -1;
    case T_Q5_B32T1: return Quantization::QuantizeQ5Row((BlockQ5_B32T1*)dst, src, cols) ? 0 : -1;
    case T_Q4_B16: return Quantization::QuantizeRow_Q4B16((BlockQ4_B16*)dst, cols / 16, src, cols) ? 0 : -1;
    case T_Q4_B32T1A: return Quantization::QuantizeRow_Q4_B32T1A((BlockQ4_B32T1*)dst, cols / 32, src, cols) ? 0 : -1;
    case T_Q4_B32T1B: return Quantization::QuantizeRow_Q4_B32T1B((BlockQ4_B32T1*)dst, cols / 32, src, cols) ? 0 : -1;
    case T_Q4_B64T1: return Quantization::QuantizeRow_Q4_B64T1((BlockQ4_B64T1*)dst, cols / 64, src, cols) ? 0 : -1;
    case T_Q3H_B64T1: return Quantization::QuantizeRow_Q3H_B64T1((BlockQ3H_B64T1*)dst, cols / 64, src, cols) ? 0 : -1;
    case T_Q3_B32T1A: return Quantization::QuantizeRow_Q3_B32T1A((BlockQ3_B32T1*)dst, cols / 32, src, cols) ? 0 : -1;
    case T_Q3_B32T1B: return Quantization::QuantizeRow_Q3_B32T1B((BlockQ3_B32T1*)dst, cols / 32, src, cols) ? 0 : -1;
    case T_Q2_B32T1A: return Quantization::QuantizeRow_Q2_B32T1A((BlockQ2_B32T1*)dst, cols / 32, src, cols) ? 0 : -1;
    case T_Q2_B32T1B: return Quantization::QuantizeRow_Q2_B32T1B((BlockQ2_B32T1*)dst, cols / 32, src, cols) ? 0 : -1;
    default: return -2;
    }
}

template <typename T>
int dequantize_row(int dtype, const uint8_t *src, int cols, T *dst)
{
    switch (dtype) {
    case T_Q8_B32T1: for (int k = 0; k < cols / 32; k++) Quantization::DequantizeQ8_B32T1(dst + 32 * k, (const BlockQ8_B32T1*)src + k); return 0;
    case T_Q8_B32T2: for (int k = 0; k < cols / 32; k++) Quantization::DequantizeQ8_B32T2(dst + 32 * k, (const BlockQ8_B32T2*)src + k); return 0;
    case T_Q6_B64T1: for (int k = 0; k < cols / 64; k++) Quantization::DequantizeQ6_B64T1(dst + 64 * k, (const BlockQ6_B64T1*)src + k); return 0;
    case T_Q5_B64T1: for (int k = 0; k < cols / 64; k++) Quantization::DequantizeQ5_B64T1(dst + 64 * k, (const BlockQ5_B64T1*)src + k); return 0;
    case T_Q5_B32T1: for (int k = 0; k < cols / 32; k++) Quantization::DequantizeQ5Block(dst + 32 * k, (const BlockQ5_B32T1*)src + k); return 0;
    case T_Q4_B16: for (int k = 0; k < cols / 16; k++) Quantization::DequantizeQ4_B16(dst + 16 * k, (const BlockQ4_B16*)src + k); return 0;
    case T_Q4_B32T1A: case T_Q4_B32T1B:
        for (int k = 0; k < cols / 32; k++) Quantization::DequantizeQ4_B32T1(dst + 32 * k, (const BlockQ4_B32T1*)src + k); return 0;
    case T_Q4_B64T1: for (int k = 0; k < cols / 64; k++) Quantization::DequantizeQ4_B64T1(dst + 64 * k, (const BlockQ4_B64T1*)src + k); return 0;
    case T_Q3H_B64T1: for (int k = 0; k < cols / 64; k++) Quantization::DequantizeQ3H_B64T1(dst + 64 * k, (const BlockQ3H_B64T1*)src + k); return 0;
    case T_Q3_B32T1A: case T_Q3_B32T1B:
        for (int k = 0; k < cols / 32; k++) Quantization::DequantizeQ3_B32T1(dst + 32 * k, (const BlockQ3_B32T1*)src + k); return 0;
    case T_Q2_B32T1A: case T_Q2_B32T1B:
        for (int k = 0; k < cols / 32; k++) Quantization::DequantizeQ2_B32T1(dst + 32 * k, (const BlockQ2_B32T1*)src + k); return 0;
    default: return -2;
    }
}
} // namespace

extern "C" {

int ref_block_bytes(int dtype)
{
    switch (dtype) {
    case T_Q8_B32T1: return (int)sizeof(BlockQ8_B32T1);
    case T_Q8_B32T2: return (int)sizeof(BlockQ8_B32T2);
    case T_Q6_B64T1: return (int)sizeof(BlockQ6_B64T1);
    case T_Q5_B64T1: return (int)sizeof(BlockQ5_B64T1);
    case T_Q5_B32T1: return (int)sizeof(BlockQ5_B32T1);
    case T_Q4_B16: return (int)sizeof(BlockQ4_B16);
    case T_Q4_B32T1A: case T_Q4_B32T1B: return (int)sizeof(BlockQ4_B32T1);
    case T_Q4_B64T1: return (int)sizeof(BlockQ4_B64T1);
    case T_Q3H_B64T1: return (int)sizeof(BlockQ3H_B64T1);
    case T_Q3_B32T1A: case T_Q3_B32T1B: return (int)sizeof(BlockQ3_B32T1);
    case T_Q2_B32T1A: case T_Q2_B32T1B: return (int)sizeof(BlockQ2_B32T1);
    default: return 0;
    }
}

// src: F16 bit patterns [rows][cols]; dst: packed blocks, row stride cols/cap*block_bytes.
int ref_quantize_rows_f16(int dtype, const uint16_t *src, int rows, int cols, uint8_t *dst, int row_bytes)
{
    static_assert(sizeof(inferflow_fp16) == 2, "fp16 size");
    for (int r = 0; r < rows; r++) {
        const inferflow_fp16 *s = reinterpret_cast<const inferflow_fp16*>(src) + (size_t)r * cols;
        int rc = quantize_row<inferflow_fp16>(dtype, s, cols, dst + (size_t)r * row_bytes);
        if (rc != 0) return rc;
    }
    return 0;
}

int ref_quantize_rows_f32(int dtype, const float *src, int rows, int cols, uint8_t *dst, int row_bytes)
{
    for (int r = 0; r < rows; r++) {
        int rc = quantize_row<float>(dtype, src + (size_t)r * cols, cols, dst + (size_t)r * row_bytes);
        if (rc != 0) return rc;
    }
    return 0;
}

int ref_dequantize_rows_f16(int dtype, const uint8_t *src, int rows, int cols, uint16_t *dst, int row_bytes)
{
    for (int r = 0; r < rows; r++) {
        inferflow_fp16 *d = reinterpret_cast<inferflow_fp16*>(dst) + (size_t)r * cols;
        int rc = dequantize_row<inferflow_fp16>(dtype, src + (size_t)r * row_bytes, cols, d);
        if (rc != 0) return rc;
    }
    return 0;
}

int ref_dequantize_rows_f32(int dtype, const uint8_t *src, int rows, int cols, float *dst, int row_bytes)
{
    for (int r = 0; r < rows; r++) {
        int rc = dequantize_row<float>(dtype, src + (size_t)r * row_bytes, cols, dst + (size_t)r * cols);
        if (rc != 0) return rc;
    }
    return 0;
}

// GetInt4 words for every 4-element group of every block of a row set.
// out: int32 [rows][cols/4].  Only the formats that have a GetInt4 overload.
int ref_get_int4_rows(int dtype, const uint8_t *src, int rows, int cols, int32_t *out, int row_bytes)
{
    for (int r = 0; r < rows; r++) {
        const uint8_t *row = src + (size_t)r * row_bytes;
        int32_t *o = out + (size_t)r * (cols / 4);
        switch (dtype) {
        case T_Q8_B32T2: for (int k = 0; k < cols / 32; k++) for (int g = 0; g < 8; g++) o[k * 8 + g] = Quantization::GetInt4(((const BlockQ8_B32T2*)row)[k], 4 * g); break;
        case T_Q6_B64T1: for (int k = 0; k < cols / 64; k++) for (int g = 0; g < 16; g++) o[k * 16 + g] = Quantization::GetInt4(((const BlockQ6_B64T1*)row)[k], 4 * g); break;
        case T_Q5_B64T1: for (int k = 0; k < cols / 64; k++) for (int g = 0; g < 16; g++) o[k * 16 + g] = Quantization::GetInt4(((const BlockQ5_B64T1*)row)[k], 4 * g); break;
        case T_Q4_B32T1A: case T_Q4_B32T1B:
            for (int k = 0; k < cols / 32; k++) for (int g = 0; g < 8; g++) o[k * 8 + g] = Quantization::GetInt4(((const BlockQ4_B32T1*)row)[k], 4 * g); break;
        case T_Q4_B64T1: for (int k = 0; k < cols / 64; k++) for (int g = 0; g < 16; g++) o[k * 16 + g] = Quantization::GetInt4(((const BlockQ4_B64T1*)row)[k], 4 * g); break;
        case T_Q3H_B64T1: for (int k = 0; k < cols / 64; k++) for (int g = 0; g < 16; g++) o[k * 16 + g] = Quantization::GetInt4(((const BlockQ3H_B64T1*)row)[k], 4 * g); break;
        default: return -2;
        }
    }
    return 0;
}

// fp16 <-> fp32 of the reference's host half type (pins orc_f2h / orc_h2f).
void ref_f2h(const float *src, uint16_t *dst, int n)
{
    for (int i = 0; i < n; i++) { inferflow_fp16 h = (inferflow_fp16)src[i]; std::memcpy(dst + i, &h, 2); }
}
void ref_h2f(const uint16_t *src, float *dst, int n)
{
    for (int i = 0; i < n; i++) { inferflow_fp16 h; std::memcpy(&h, src + i, 2); dst[i] = (float)h; }
}

} // extern "C"
