// ifa_decode_formats.h -- per-format register images for the fused decode GEMV
// (k_dec_gemv, ifa_decode_kernels.h): the activation slice a lane keeps (XRegs*) and one
// weight row's slice (WRow*), for every weight format the reference's int8-activation
// GEMV accepts (GetUseFullQuantGemv, src/transformer/inference_worker.cc:2707-2730):
//   Q4_B32T1A/B  Q8_B32T2  Q4_B64T1  Q3H_B64T1  Q5_B64T1  Q6_B64T1
//
// Rows are read from the row-local plane layout (ifa_tiled.h): lane l owns weight blocks
// l+64j.  Codes are expanded to int8x4 words with packed integer tricks (no per-element
// extraction), multiplied with v_dot4_i32_i8 against the int8 activation, and each
// 32-element half block contributes   xs * (dot*scale + xsum*base)   in the same order as
// the op-level kernel (ax8_term, ifa_gemv.hip), so fused and op-by-op results are bit-identical.
//
// Code order inside a byte plane is always "byte i = elements 2i (low) and 2i+1 (high)", so the
// activation is kept split into even / odd elements (xe / xo); Q8_B32T2 uses natural order.
#pragma once
#include "ifa_device.h"
#include "ifa_tiled.h"

namespace ifa {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct XLds;   // ifa_decode_kernels.h

template <typename T>
__device__ __forceinline__ T nt_load(const void *p) { return __builtin_nontemporal_load(reinterpret_cast<const T *>(p)); }

// one half block's fp32 contribution (same expression as ax8_term<DT>)
__device__ __forceinline__ float dec_term(int d, float scale, float base, float xsum, float xs)
{
    float t = (float)d * scale;
    float u = xsum * base;
    t = t + u;
    return xs * t;
}

__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return a * b; }   // operands < 2^24: compiles to v_mul_u32_u24

// ---- activation registers -------------------------------------------------
// 64-element weight blocks: x blocks 2*blk and 2*blk+1, even/odd split
template <int NJ>
struct XRegsB64 {
    int xe[NJ][2][4], xo[NJ][2][4];
    float xs[NJ][2], xsf[NJ][2];
    __device__ __forceinline__ void load(const int8_t *codes, const float *scale, const float *xsum, int lane, int nblk)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = lane + 64 * j;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                xs[j][h] = 0.0f; xsf[j][h] = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; w++) { xe[j][h][w] = 0; xo[j][h][w] = 0; }
            }
            if (blk < nblk) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int xb = 2 * blk + h;
                    const u32x4 a = *reinterpret_cast<const u32x4 *>(codes + (size_t)xb * 32);
                    const u32x4 b = *reinterpret_cast<const u32x4 *>(codes + (size_t)xb * 32 + 16);
                    const uint32_t d[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        xe[j][h][w] = (int)__builtin_amdgcn_perm(d[2 * w + 1], d[2 * w], 0x06040200u);
                        xo[j][h][w] = (int)__builtin_amdgcn_perm(d[2 * w + 1], d[2 * w], 0x07050301u);
                    }
                    xs[j][h] = scale[xb];
                    xsf[j][h] = xsum[xb];
                }
            }
        }
    }
};

// 32-element blocks, natural element order, no base term (Q8_B32T2 weights)
template <int NJ>
struct XRegsNat {
    int xn[NJ][8];
    float xs[NJ];
    __device__ __forceinline__ void load(const int8_t *codes, const float *scale, const float *, int lane, int nblk)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = lane + 64 * j;
            xs[j] = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; w++) xn[j][w] = 0;
            if (blk < nblk) {
                const u32x4 a = *reinterpret_cast<const u32x4 *>(codes + (size_t)blk * 32);
                const u32x4 b = *reinterpret_cast<const u32x4 *>(codes + (size_t)blk * 32 + 16);
#pragma unroll
                for (int w = 0; w < 4; w++) { xn[j][w] = (int)a[w]; xn[j][4 + w] = (int)b[w]; }
                xs[j] = scale[blk];
            }
        }
    }
};

// ---- weight rows ----------------------------------------------------------
// All loads are unconditional with the block index clamped into the row (see WRowQ4).

// Q8_B32T2 {scale f16, int8 data[32]}            tiled: [data32][scale2]
template <int NJ>
struct WRowQ8T2 {
    u32x4 c0[NJ], c1[NJ];
    uint16_t sc[NJ];
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(lane + 64 * j, nblk - 1);
            c0[j] = nt_load<u32x4>(wrow + (size_t)blk * 32);
            c1[j] = nt_load<u32x4>(wrow + (size_t)blk * 32 + 16);
            sc[j] = nt_load<uint16_t>(wrow + (size_t)nblk * 32 + (size_t)blk * 2);
        }
    }
    __device__ __forceinline__ float dot(const XRegsNat<NJ> &X) const
    {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            int d = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                d = sdot4((int)c0[j][w], X.xn[j][w], d);
                d = sdot4((int)c1[j][w], X.xn[j][4 + w], d);
            }
            float t = (float)d * hbits2f(sc[j]);
            acc = acc + X.xs[j] * t;
        }
        return acc;
    }
};

// Q4_B64T1 {base, scale, nibbles[32]}             tiled: [data32][base,scale]
template <int NJ>
struct WRowQ4B64 {
    u32x4 c[NJ][2];
    uint32_t sb[NJ];
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(lane + 64 * j, nblk - 1);
            c[j][0] = nt_load<u32x4>(wrow + (size_t)blk * 32);
            c[j][1] = nt_load<u32x4>(wrow + (size_t)blk * 32 + 16);
            sb[j] = nt_load<uint32_t>(wrow + (size_t)nblk * 32 + (size_t)blk * 4);
        }
    }
    __device__ __forceinline__ float dot(const XRegsB64<NJ> &X) const
    {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const float base = hbits2f((uint16_t)(sb[j] & 0xFFFFu)), scale = hbits2f((uint16_t)(sb[j] >> 16));
#pragma unroll
            for (int h = 0; h < 2; h++) {
                int d = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t cw = c[j][h][w];
                    d = sdot4((int)(cw & 0x0F0F0F0Fu), X.xe[j][h][w], d);
                    d = sdot4((int)((cw >> 4) & 0x0F0F0F0Fu), X.xo[j][h][w], d);
                }
                acc = acc + dec_term(d, scale, base, X.xsf[j][h], X.xs[j][h]);
            }
        }
        return acc;
    }
};

// Q5_B64T1 {base, scale, data_h[8], nibbles[32]}   tiled: [data32][data_h8][base,scale]
// bit k of data_h byte W is the 5th bit of element 8W+k  (quantization.h:414-443)
template <int NJ>
struct WRowQ5B64 {
    u32x4 c[NJ][2];
    u32x2 hb[NJ];
    uint32_t sb[NJ];
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(lane + 64 * j, nblk - 1);
            c[j][0] = nt_load<u32x4>(wrow + (size_t)blk * 32);
            c[j][1] = nt_load<u32x4>(wrow + (size_t)blk * 32 + 16);
            hb[j] = nt_load<u32x2>(wrow + (size_t)nblk * 32 + (size_t)blk * 8);
            sb[j] = nt_load<uint32_t>(wrow + (size_t)nblk * 40 + (size_t)blk * 4);
        }
    }
    __device__ __forceinline__ float dot(const XRegsB64<NJ> &X) const
    {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const float base = hbits2f((uint16_t)(sb[j] & 0xFFFFu)), scale = hbits2f((uint16_t)(sb[j] >> 16));
#pragma unroll
            for (int h = 0; h < 2; h++) {
                int d = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t cw = c[j][h][w];
                    const uint32_t hbyte = (hb[j][h] >> (8 * w)) & 0xFFu;
                    // bits 0,2,4,6 -> bit 4 of bytes 0..3 (shifts 4,10,16,22; no colliding partial products)
                    const uint32_t he = mul24(hbyte & 0x55u, 0x410410u) & 0x10101010u;
                    const uint32_t ho = mul24((hbyte >> 1) & 0x55u, 0x410410u) & 0x10101010u;
                    d = sdot4((int)((cw & 0x0F0F0F0Fu) | he), X.xe[j][h][w], d);
                    d = sdot4((int)(((cw >> 4) & 0x0F0F0F0Fu) | ho), X.xo[j][h][w], d);
                }
                acc = acc + dec_term(d, scale, base, X.xsf[j][h], X.xs[j][h]);
            }
        }
        return acc;
    }
};

// Q6_B64T1 {base, scale, data_h[16], nibbles[32]}  tiled: [data32][data_h16][base,scale]
// bits (2k,2k+1) of the 16-bit word data_h[2W..2W+1] are bits 4-5 of element 8W+k  (:240-266)
template <int NJ>
struct WRowQ6B64 {
    u32x4 c[NJ][2];
    u32x4 hb[NJ];
    uint32_t sb[NJ];
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(lane + 64 * j, nblk - 1);
            c[j][0] = nt_load<u32x4>(wrow + (size_t)blk * 32);
            c[j][1] = nt_load<u32x4>(wrow + (size_t)blk * 32 + 16);
            hb[j] = nt_load<u32x4>(wrow + (size_t)nblk * 32 + (size_t)blk * 16);
            sb[j] = nt_load<uint32_t>(wrow + (size_t)nblk * 48 + (size_t)blk * 4);
        }
    }
    // bit pairs at 0,4,8,12 of t -> bits 4-5 of bytes 0..3
    static __device__ __forceinline__ uint32_t spread2(uint32_t t)
    {
        t &= 0x3333u;
        const uint32_t u = (t & 0x0033u) | ((t & 0x3300u) << 8);
        return ((u & 0x00030003u) << 4) | ((u & 0x00300030u) << 8);
    }
    __device__ __forceinline__ float dot(const XRegsB64<NJ> &X) const
    {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const float base = hbits2f((uint16_t)(sb[j] & 0xFFFFu)), scale = hbits2f((uint16_t)(sb[j] >> 16));
#pragma unroll
            for (int h = 0; h < 2; h++) {
                int d = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const int W = 4 * h + w;
                    const uint32_t cw = c[j][h][w];
                    const uint32_t h16 = (hb[j][W >> 1] >> (16 * (W & 1))) & 0xFFFFu;
                    d = sdot4((int)((cw & 0x0F0F0F0Fu) | spread2(h16)), X.xe[j][h][w], d);
                    d = sdot4((int)(((cw >> 4) & 0x0F0F0F0Fu) | spread2(h16 >> 2)), X.xo[j][h][w], d);
                }
                acc = acc + dec_term(d, scale, base, X.xsf[j][h], X.xs[j][h]);
            }
        }
        return acc;
    }
};

// Q3H_B64T1 {base, scale, data_h[4], data_m[8], data[16]}: 32 seven-bit pair codes p = q0 + 11*q1
// (elements 2k, 2k+1; quantization.h:823-851).            tiled: [data16][data_m8][base,scale,data_h4]
// pair k: low nibble k of data, bits (2k,2k+1) of data_m, bit k of data_h.
// q1 = p/11 = (93*p + 64) >> 10 for 0 <= p < 128 (two pairs per 32-bit multiply-add);
// q0 = p - 11*q1 is never formed:  sum q0*xe + q1*xo = sum p*xe + q1*xo - 11 * sum q1*xe.
template <int NJ>
struct WRowQ3H {
    u32x4 c[NJ];
    u32x2 m[NJ];
    u32x2 sbh[NJ];   // [0] = base | scale<<16, [1] = data_h bits
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(lane + 64 * j, nblk - 1);
            c[j] = nt_load<u32x4>(wrow + (size_t)blk * 16);
            m[j] = nt_load<u32x2>(wrow + (size_t)nblk * 16 + (size_t)blk * 8);
            sbh[j] = nt_load<u32x2>(wrow + (size_t)nblk * 24 + (size_t)blk * 8);
        }
    }
    // four pair codes (one per byte) from 16 bits of nibbles, one data_m byte and 4 data_h bits
    static __device__ __forceinline__ uint32_t pairs4(uint32_t nib16, uint32_t m8, uint32_t h4)
    {
        uint32_t t = (nib16 | (nib16 << 8)) & 0x00FF00FFu;
        t = (t | (t << 4)) & 0x0F0F0F0Fu;
        const uint32_t mb = (mul24(m8 & 0x33u, 0x41041u) & 0x00030003u) | (mul24(m8 & 0xCCu, 0x41041u) & 0x03000300u);
        const uint32_t hbits = mul24(h4, 0x204081u) & 0x01010101u;
        return t | (mb << 4) | (hbits << 6);
    }
    static __device__ __forceinline__ uint32_t div11x4(uint32_t p)
    {
        const uint32_t a = p & 0x00FF00FFu, b = (p >> 8) & 0x00FF00FFu;
        const uint32_t qa = ((a * 93u + 0x00400040u) >> 10) & 0x000F000Fu;
        const uint32_t qb = ((b * 93u + 0x00400040u) >> 10) & 0x000F000Fu;
        return qa | (qb << 8);
    }
    __device__ __forceinline__ float dot(const XRegsB64<NJ> &X) const
    {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const float base = hbits2f((uint16_t)(sbh[j][0] & 0xFFFFu)), scale = hbits2f((uint16_t)(sbh[j][0] >> 16));
            const uint32_t H = sbh[j][1];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                int da = 0, db = 0;
#pragma unroll
                for (int w2 = 0; w2 < 2; w2++) {            // data dword W = 2h+w2 holds pairs 8W..8W+7
                    const int W = 2 * h + w2;
                    const uint32_t cw = c[j][W];
                    const uint32_t mm = (m[j][W >> 1] >> (16 * (W & 1))) & 0xFFFFu;
                    const uint32_t hh = (H >> (8 * W)) & 0xFFu;
#pragma unroll
                    for (int half = 0; half < 2; half++) {  // pairs 8W+4*half .. +3 = x-block elements 16*w2+8*half .. +7
                        const uint32_t P = pairs4((cw >> (16 * half)) & 0xFFFFu, (mm >> (8 * half)) & 0xFFu, (hh >> (4 * half)) & 0xFu);
                        const uint32_t Q1 = div11x4(P);
                        const int xw = 2 * w2 + half;
                        da = sdot4((int)P, X.xe[j][h][xw], da);
                        da = sdot4((int)Q1, X.xo[j][h][xw], da);
                        db = sdot4((int)Q1, X.xe[j][h][xw], db);
                    }
                }
                acc = acc + dec_term(da - 11 * db, scale, base, X.xsf[j][h], X.xs[j][h]);
            }
        }
        return acc;
    }
};

} // namespace ifa
