// ifa_codec.h -- block codecs on the device, reference byte layout (AoS).
//
// Formats: src/common/quant_types.h:11-174.  Quantizers and decoders follow
// src/common/quantization.h (line ranges per function below) and are checked
// bit-for-bit against oracle/ (which is pinned to the reference header).
// All fp32 expressions are contraction-free (the library is built with
// -ffp-contract=off) so they round exactly like the host-compiled reference.
#pragma once
#include "ifa_device.h"

namespace ifa {

// A block copied to registers as 16-bit words (every block size is even and
// every block address is at least 2-byte aligned).
template <int BYTES>
struct RawBlock {
    uint16_t w[BYTES / 2];
    __device__ __forceinline__ void load(const uint8_t *p)
    {
        const uint16_t *q = reinterpret_cast<const uint16_t *>(p);
#pragma unroll
        for (int i = 0; i < BYTES / 2; i++) w[i] = q[i];
    }
    // the same from a pointer known to be DEVICE memory (global loads even when the pointer's origin is opaque to the compiler:
    // FLAT loads also count on lgkmcnt, so LDS waits would wait for them)
    __device__ __forceinline__ void load_global(const uint8_t *p)
    {
        typedef const __attribute__((address_space(1))) uint16_t gu16;
        gu16 *q = (gu16 *)p;
#pragma unroll
        for (int i = 0; i < BYTES / 2; i++) w[i] = q[i];
    }
    __device__ __forceinline__ void store(uint8_t *p) const
    {
        uint16_t *q = reinterpret_cast<uint16_t *>(p);
#pragma unroll
        for (int i = 0; i < BYTES / 2; i++) q[i] = w[i];
    }
    __device__ __forceinline__ uint32_t u8(int i) const { return (w[i >> 1] >> ((i & 1) * 8)) & 0xFFu; }
    __device__ __forceinline__ uint32_t u16(int i) const { return w[i >> 1]; }  // i even
    __device__ __forceinline__ void set_u8(int i, uint32_t v)
    {
        if (i & 1) w[i >> 1] = (uint16_t)((w[i >> 1] & 0x00FFu) | ((v & 0xFFu) << 8));
        else w[i >> 1] = (uint16_t)((w[i >> 1] & 0xFF00u) | (v & 0xFFu));
    }
    __device__ __forceinline__ void set_u16(int i, uint32_t v) { w[i >> 1] = (uint16_t)v; }
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < BYTES / 2; i++) w[i] = 0;
    }
};

// --------------------------------------------------------------------------
// decode: integer codes in element order + fp32 scale/base,
// value(i) = q[i]*scale + base   (Q8_B32T2: q[i]*scale)
// --------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ void decode_block(const RawBlock<block_bytes(DT)> &b, int *q, float &scale, float &base)
{
    if constexpr (DT == Q8_B32T1) {            // quantization.h:94-108
        base = hbits2f(b.u16(0)); scale = hbits2f(b.u16(2));
#pragma unroll
        for (int i = 0; i < 32; i++) q[i] = (int)b.u8(4 + i);
    } else if constexpr (DT == Q8_B32T2) {     // :171-184
        scale = hbits2f(b.u16(0)); base = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; i++) q[i] = (int)(int8_t)b.u8(2 + i);
    } else if constexpr (DT == Q6_B64T1) {     // :240-266
        base = hbits2f(b.u16(0)); scale = hbits2f(b.u16(2));
#pragma unroll
        for (int idx = 0; idx < 16; idx++) {
            uint32_t qh = b.u8(4 + idx);
            uint32_t qd = b.u16(20 + 2 * idx);
            q[4 * idx] = (int)((qd & 0x0F) | ((qh & 0x03) << 4));
            q[4 * idx + 1] = (int)(((qd >> 4) & 0x0F) | (((qh >> 2) & 0x03) << 4));
            q[4 * idx + 2] = (int)(((qd >> 8) & 0x0F) | (((qh >> 4) & 0x03) << 4));
            q[4 * idx + 3] = (int)(((qd >> 12) & 0x0F) | (((qh >> 6) & 0x03) << 4));
        }
    } else if constexpr (DT == Q5_B64T1) {     // :414-443
        base = hbits2f(b.u16(0)); scale = hbits2f(b.u16(2));
#pragma unroll
        for (int idx = 0; idx < 16; idx++) {
            uint32_t qh = b.u8(4 + idx / 2);
            if (idx % 2 != 0) qh >>= 4;
            uint32_t qd = b.u16(12 + 2 * idx);
            q[4 * idx] = (int)((qd & 0x0F) | ((qh & 0x01) << 4));
            q[4 * idx + 1] = (int)(((qd >> 4) & 0x0F) | (((qh >> 1) & 0x01) << 4));
            q[4 * idx + 2] = (int)(((qd >> 8) & 0x0F) | (((qh >> 2) & 0x01) << 4));
            q[4 * idx + 3] = (int)(((qd >> 12) & 0x0F) | (((qh >> 3) & 0x01) << 4));
        }
    } else if constexpr (DT == Q5_B32T1) {     // :325-345 (struct order: scale, base)
        scale = hbits2f(b.u16(0)); base = hbits2f(b.u16(2));
        uint32_t qh = b.u16(4) | (b.u16(6) << 16);
#pragma unroll
        for (int idx = 0; idx < 16; idx++) {
            uint32_t d = b.u8(8 + idx);
            q[idx] = (int)((d & 0x0F) | (((qh >> idx) & 1) << 4));
            q[idx + 16] = (int)((d >> 4) | (((qh >> (idx + 16)) & 1) << 4));
        }
    } else if constexpr (DT == Q4_B16) {       // :638-655, :41-59
        base = (float)(int)b.u8(0) / 100.0f - 1.0f;
        scale = (float)b.u8(1) / 1000;
#pragma unroll
        for (int i = 0; i < 8; i++) { uint32_t d = b.u8(2 + i); q[2 * i] = (int)(d & 0x0F); q[2 * i + 1] = (int)(d >> 4); }
    } else if constexpr (DT == Q4_B32T1A || DT == Q4_B32T1B) {   // :516-533
        base = hbits2f(b.u16(0)); scale = hbits2f(b.u16(2));
#pragma unroll
        for (int i = 0; i < 16; i++) { uint32_t d = b.u8(4 + i); q[2 * i] = (int)(d & 0x0F); q[2 * i + 1] = (int)(d >> 4); }
    } else if constexpr (DT == Q4_B64T1) {     // :735-754
        base = hbits2f(b.u16(0)); scale = hbits2f(b.u16(2));
#pragma unroll
        for (int i = 0; i < 32; i++) { uint32_t d = b.u8(4 + i); q[2 * i] = (int)(d & 0x0F); q[2 * i + 1] = (int)(d >> 4); }
    } else if constexpr (DT == Q3H_B64T1) {    // :823-851
        base = hbits2f(b.u16(0)); scale = hbits2f(b.u16(2));
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            uint32_t u16v = b.u16(16 + 2 * idx);
            uint32_t m8 = b.u8(8 + idx);
            uint32_t hb = b.u8(4 + idx / 2);
            uint32_t h8 = (idx % 2 == 0) ? (hb & 0x0F) : ((hb & 0xF0) >> 4);
            uint32_t p0 = ((u16v & 0x000F)) | ((m8 & 0x03) << 4) | ((h8 & 0x01) << 6);
            uint32_t p1 = ((u16v & 0x00F0) >> 4) | ((m8 & 0x0C) << 2) | ((h8 & 0x02) << 5);
            uint32_t p2 = ((u16v & 0x0F00) >> 8) | ((m8 & 0x30)) | ((h8 & 0x04) << 4);
            uint32_t p3 = ((u16v & 0xF000) >> 12) | ((m8 & 0xC0) >> 2) | ((h8 & 0x08) << 3);
            q[8 * idx] = (int)(p0 % 11); q[8 * idx + 1] = (int)(p0 / 11);
            q[8 * idx + 2] = (int)(p1 % 11); q[8 * idx + 3] = (int)(p1 / 11);
            q[8 * idx + 4] = (int)(p2 % 11); q[8 * idx + 5] = (int)(p2 / 11);
            q[8 * idx + 6] = (int)(p3 % 11); q[8 * idx + 7] = (int)(p3 / 11);
        }
    } else if constexpr (DT == Q3_B32T1A || DT == Q3_B32T1B) {   // :933-961
        base = hbits2f(b.u16(0)); scale = hbits2f(b.u16(2));
#pragma unroll
        for (int idx = 0; idx < 4; idx++) {
            uint32_t u16v = b.u16(8 + 2 * idx);
            uint32_t h8 = b.u8(4 + idx);
#pragma unroll
            for (int i = 0; i < 8; i++) q[8 * idx + i] = (int)(((u16v >> (2 * i)) & 3) | (((h8 >> i) & 1) << 2));
        }
    } else if constexpr (DT == Q2_B32T1A || DT == Q2_B32T1B) {   // :1075-1095
        base = hbits2f(b.u16(0)); scale = hbits2f(b.u16(2));
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t d = b.u8(4 + i);
            q[4 * i] = (int)(d & 3); q[4 * i + 1] = (int)((d >> 2) & 3);
            q[4 * i + 2] = (int)((d >> 4) & 3); q[4 * i + 3] = (int)(d >> 6);
        }
    }
}

template <int DT>
__device__ __forceinline__ float block_value(int q, float scale, float base)
{
    if constexpr (DT == Q8_B32T2) return (float)q * scale;
    float t = (float)q * scale;
    return t + base;
}

// --------------------------------------------------------------------------
// quantize one block of fp32 values (already widened from the source type)
// --------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void value_range(const float *s, float &mn, float &mx)
{   // Quantization::GetValueRange, quantization.h:68-85
    mn = s[0]; mx = s[0];
#pragma unroll
    for (int i = 1; i < N; i++) {
        float v = s[i];
        if (mn > v) mn = v;
        if (mx < v) mx = v;
    }
}

__device__ __forceinline__ float inv_scale_of(float scale) { return scale >= 0.00001f ? (1.0f / scale) : 0.0f; }
__device__ __forceinline__ uint32_t qcode(float v, float mn, float inv, float radd)
{
    float t = (v - mn) * inv;
    return (uint32_t)(t + radd);
}

template <int DT>
__device__ __forceinline__ void quantize_block(const float *s, RawBlock<block_bytes(DT)> &b)
{
    b.clear();
    if constexpr (DT == Q8_B32T1) {            // quantization.h:110-152
        float mn, mx; value_range<32>(s, mn, mx);
        float scale = (mx - mn) / 255;
        float inv = inv_scale_of(scale);
        b.set_u16(0, f2hbits(mn)); b.set_u16(2, f2hbits(scale));
#pragma unroll
        for (int r = 0; r < 32; r++) { uint32_t q = qcode(s[r], mn, inv, 0.5f); q = q > 255 ? 255 : q; b.set_u8(4 + r, q); }
    } else if constexpr (DT == Q8_B32T2) {     // device alg-2: src/kernels/tensor_quant.h:44-82
        float mxv = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; r++) mxv = fmaxf(mxv, fabsf(s[r]));
        float scale = mxv / 127;
#pragma unroll
        for (int r = 0; r < 32; r++) {
            int q = scale <= 0.000001f ? 0 : (int)roundf(s[r] / scale);
            q = min(max(q, -128), 127);
            b.set_u8(2 + r, (uint32_t)q & 0xFFu);
        }
        b.set_u16(0, f2hbits(scale));
    } else if constexpr (DT == Q6_B64T1) {     // :268-322 (divides by 62)
        float mn, mx; value_range<64>(s, mn, mx);
        float scale = (mx - mn) / 62;
        float inv = inv_scale_of(scale);
        b.set_u16(0, f2hbits(mn)); b.set_u16(2, f2hbits(scale));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            uint32_t q0 = min(qcode(s[4 * r], mn, inv, 0.5f), 63u), q1 = min(qcode(s[4 * r + 1], mn, inv, 0.5f), 63u);
            uint32_t q2 = min(qcode(s[4 * r + 2], mn, inv, 0.5f), 63u), q3 = min(qcode(s[4 * r + 3], mn, inv, 0.5f), 63u);
            b.set_u8(4 + r, (q0 >> 4) | ((q1 & 0x30) >> 2) | (q2 & 0x30) | ((q3 & 0x30) << 2));
            b.set_u8(20 + 2 * r, (q0 & 0x0F) | ((q1 & 0x0F) << 4));
            b.set_u8(20 + 2 * r + 1, (q2 & 0x0F) | ((q3 & 0x0F) << 4));
        }
    } else if constexpr (DT == Q5_B64T1) {     // :446-503 (divides by 30)
        float mn, mx; value_range<64>(s, mn, mx);
        float scale = (mx - mn) / 30;
        float inv = inv_scale_of(scale);
        b.set_u16(0, f2hbits(mn)); b.set_u16(2, f2hbits(scale));
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t q[8];
#pragma unroll
            for (int li = 0; li < 8; li++) q[li] = min(qcode(s[8 * r + li], mn, inv, 0.5f), 31u);
            b.set_u8(4 + r, ((q[0] & 0x10) >> 4) | ((q[1] & 0x10) >> 3) | ((q[2] & 0x10) >> 2) | ((q[3] & 0x10) >> 1)
                | ((q[4] & 0x10)) | ((q[5] & 0x10) << 1) | ((q[6] & 0x10) << 2) | ((q[7] & 0x10) << 3));
#pragma unroll
            for (int p = 0; p < 4; p++) b.set_u8(12 + 4 * r + p, (q[2 * p] & 0x0F) | ((q[2 * p + 1] & 0x0F) << 4));
        }
    } else if constexpr (DT == Q5_B32T1) {     // QuantizeQ5Row :348-393
        float mn, mx; value_range<32>(s, mn, mx);
        float delta = (mx - mn) / 31;
        float inv = inv_scale_of(delta);
        b.set_u16(0, f2hbits(delta)); b.set_u16(2, f2hbits(mn));
        uint32_t qh = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            uint32_t q1 = qcode(s[r], mn, inv, 0.5f), q2 = qcode(s[r + 16], mn, inv, 0.5f);
            b.set_u8(8 + r, (q1 & 0x0F) | ((q2 & 0x0F) << 4));
            qh |= (((q1 & 0x10) >> 4) << r);
            qh |= (((q2 & 0x10) >> 4) << (r + 16));
        }
        b.set_u16(4, qh & 0xFFFFu); b.set_u16(6, qh >> 16);
    } else if constexpr (DT == Q4_B16) {       // :657-712
        float mn, mx; value_range<16>(s, mn, mx);
        {   // AdjustBase :61-65 (double arithmetic on the +100.01)
            float b100 = mn * 100;
            uint32_t u8 = (uint32_t)(int)((double)b100 + 100.01) & 0xFFu;
            mn = (float)(int)u8 / 100.0f - 1.0f;
        }
        float scale = (mx - mn) / 15;
        float inv = inv_scale_of(scale);
        { float t = mn * 100; b.set_u8(0, (uint32_t)(int)(t + 100.5f) & 0xFFu); }
        { float t = scale * 1000; b.set_u8(1, (uint32_t)(int)(t + 0.5f) & 0xFFu); }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t q1 = min(qcode(s[2 * r], mn, inv, 0.5f), 15u), q2 = min(qcode(s[2 * r + 1], mn, inv, 0.5f), 15u);
            b.set_u8(2 + r, (q1 & 0x0F) | ((q2 & 0x0F) << 4));
        }
    } else if constexpr (DT == Q4_B32T1A || DT == Q4_B32T1B) {   // :535-586, :589-632
        constexpr bool B = (DT == Q4_B32T1B);
        float mn, mx; value_range<32>(s, mn, mx);
        float scale = B ? (mx - mn) / 16 : (mx - mn) / 15;
        float inv = inv_scale_of(scale);
        float base = mn;
        if constexpr (B) { float hs = 0.5f * scale; base = mn + hs; }
        b.set_u16(0, f2hbits(base)); b.set_u16(2, f2hbits(scale));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            uint32_t q1 = min(qcode(s[2 * r], mn, inv, B ? 0.0001f : 0.5f), 15u);
            uint32_t q2 = min(qcode(s[2 * r + 1], mn, inv, B ? 0.0001f : 0.5f), 15u);
            b.set_u8(4 + r, (q1 & 0x0F) | ((q2 & 0x0F) << 4));
        }
    } else if constexpr (DT == Q4_B64T1) {     // :757-803 (divides by 14)
        float mn, mx; value_range<64>(s, mn, mx);
        float scale = (mx - mn) / 14;
        float inv = inv_scale_of(scale);
        b.set_u16(0, f2hbits(mn)); b.set_u16(2, f2hbits(scale));
#pragma unroll
        for (int r = 0; r < 32; r++) {
            uint32_t q0 = min(qcode(s[2 * r], mn, inv, 0.5f), 15u), q1 = min(qcode(s[2 * r + 1], mn, inv, 0.5f), 15u);
            b.set_u8(4 + r, q0 | (q1 << 4));
        }
    } else if constexpr (DT == Q3H_B64T1) {    // :854-926 (11 levels, base-11 pairs)
        float mn, mx; value_range<64>(s, mn, mx);
        float scale = (mx - mn) / 10;
        float inv = inv_scale_of(scale);
        b.set_u16(0, f2hbits(mn)); b.set_u16(2, f2hbits(scale));
        uint32_t data_h = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            int qa[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float t = (s[8 * r + i] - mn) * inv;
                int q = (int)(t + 0.5f);
                qa[i] = min(max(q, 0), 10);
            }
            uint32_t q1 = (uint32_t)(qa[0] + qa[1] * 11), q2 = (uint32_t)(qa[2] + qa[3] * 11);
            uint32_t q3 = (uint32_t)(qa[4] + qa[5] * 11), q4 = (uint32_t)(qa[6] + qa[7] * 11);
            b.set_u8(16 + 2 * r, (q1 & 0x0F) | ((q2 & 0x0F) << 4));
            b.set_u8(16 + 2 * r + 1, (q3 & 0x0F) | ((q4 & 0x0F) << 4));
            b.set_u8(8 + r, ((q1 & 0x30) >> 4) | ((q2 & 0x30) >> 2) | (q3 & 0x30) | ((q4 & 0x30) << 2));
            if (r % 2 == 0) {
                data_h = ((q1 & 0x40) >> 6) | ((q2 & 0x40) >> 5) | ((q3 & 0x40) >> 4) | ((q4 & 0x40) >> 3);
            } else {
                data_h = data_h | ((q1 & 0x40) >> 2) | ((q2 & 0x40) >> 1) | (q3 & 0x40) | ((q4 & 0x40) << 1);
                b.set_u8(4 + r / 2, data_h);
                data_h = 0;
            }
        }
    } else if constexpr (DT == Q3_B32T1A || DT == Q3_B32T1B) {   // :964-1014, :1017-1068
        constexpr bool B = (DT == Q3_B32T1B);
        float mn, mx; value_range<32>(s, mn, mx);
        float scale = B ? (mx - mn) / 8 : (mx - mn) / 7;
        float inv = inv_scale_of(scale);
        float base = mn;
        if constexpr (B) { float hs = 0.5f * scale; base = mn + hs; }
        b.set_u16(0, f2hbits(base)); b.set_u16(2, f2hbits(scale));
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t q[8];
#pragma unroll
            for (int i = 0; i < 8; i++) q[i] = min(qcode(s[8 * r + i], mn, inv, B ? 0.0001f : 0.5f), 7u);
            b.set_u8(8 + 2 * r, (q[0] & 3) | ((q[1] & 3) << 2) | ((q[2] & 3) << 4) | ((q[3] & 3) << 6));
            b.set_u8(8 + 2 * r + 1, (q[4] & 3) | ((q[5] & 3) << 2) | ((q[6] & 3) << 4) | ((q[7] & 3) << 6));
            b.set_u8(4 + r, ((q[0] & 4) >> 2) | ((q[1] & 4) >> 1) | (q[2] & 4) | ((q[3] & 4) << 1)
                | ((q[4] & 4) << 2) | ((q[5] & 4) << 3) | ((q[6] & 4) << 4) | ((q[7] & 4) << 5));
        }
    } else if constexpr (DT == Q2_B32T1A || DT == Q2_B32T1B) {   // :1098-1144, :1147-1193
        constexpr bool B = (DT == Q2_B32T1B);
        float mn, mx; value_range<32>(s, mn, mx);
        float scale = B ? (mx - mn) / 4 : (mx - mn) / 3;
        float inv = inv_scale_of(scale);
        float base = mn;
        if constexpr (B) { float hs = 0.5f * scale; base = mn + hs; }
        b.set_u16(0, f2hbits(base)); b.set_u16(2, f2hbits(scale));
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t q[4];
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = min(qcode(s[4 * r + i], mn, inv, B ? 0.0001f : 0.5f), 3u);
            b.set_u8(4 + r, q[0] | (q[1] << 2) | (q[2] << 4) | (q[3] << 6));
        }
    }
}

// Dispatch a functor templated on the dtype id: f.template operator()<DT>().
#define IFA_DISPATCH_QUANT_DTYPE(dt, ...)                       \
    switch (dt) {                                                \
    case ::ifa::Q8_B32T1: { constexpr int DT = ::ifa::Q8_B32T1; __VA_ARGS__; break; }   \
    case ::ifa::Q8_B32T2: { constexpr int DT = ::ifa::Q8_B32T2; __VA_ARGS__; break; }   \
    case ::ifa::Q6_B64T1: { constexpr int DT = ::ifa::Q6_B64T1; __VA_ARGS__; break; }   \
    case ::ifa::Q5_B64T1: { constexpr int DT = ::ifa::Q5_B64T1; __VA_ARGS__; break; }   \
    case ::ifa::Q5_B32T1: { constexpr int DT = ::ifa::Q5_B32T1; __VA_ARGS__; break; }   \
    case ::ifa::Q4_B16: { constexpr int DT = ::ifa::Q4_B16; __VA_ARGS__; break; }       \
    case ::ifa::Q4_B32T1A: { constexpr int DT = ::ifa::Q4_B32T1A; __VA_ARGS__; break; } \
    case ::ifa::Q4_B32T1B: { constexpr int DT = ::ifa::Q4_B32T1B; __VA_ARGS__; break; } \
    case ::ifa::Q4_B64T1: { constexpr int DT = ::ifa::Q4_B64T1; __VA_ARGS__; break; }   \
    case ::ifa::Q3H_B64T1: { constexpr int DT = ::ifa::Q3H_B64T1; __VA_ARGS__; break; } \
    case ::ifa::Q3_B32T1A: { constexpr int DT = ::ifa::Q3_B32T1A; __VA_ARGS__; break; } \
    case ::ifa::Q3_B32T1B: { constexpr int DT = ::ifa::Q3_B32T1B; __VA_ARGS__; break; } \
    case ::ifa::Q2_B32T1A: { constexpr int DT = ::ifa::Q2_B32T1A; __VA_ARGS__; break; } \
    case ::ifa::Q2_B32T1B: { constexpr int DT = ::ifa::Q2_B32T1B; __VA_ARGS__; break; } \
    default: return ifa_fail(IFA_ERR_DTYPE, "unsupported dtype %d", (int)(dt));  \
    }

} // namespace ifa
