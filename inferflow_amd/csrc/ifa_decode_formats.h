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
__device__ __forceinline__ T nt_load(const void *p)
{
    // weights live in device memory: say so.  A pointer that reached the kernel through memory (the MoE expert table) or an
    // integer would otherwise be loaded from with FLAT instructions, which count on lgkmcnt too -- every LDS wait then also waits
    // for the weight requests in flight (ifa_gemm_rows_mfma_body.h, rows-trace)
    typedef const __attribute__((address_space(1))) T gT;
    return __builtin_nontemporal_load((gT *)p);
}

// Where a weight row's bytes come from: HBM (non-temporal requests, the five-launch kernels) or the LDS ring the
// persistent layer kernel's loader wave fills (ifa_decode_persist.h; offsets wrap at the ring size).  The WRow*::load_src
// methods below take either, so both paths decode and accumulate a row with the same code.
struct WSrcGlobal {
    const uint8_t *p;
    template <typename T> __device__ __forceinline__ T ld(uint32_t off) const { return nt_load<T>(p + off); }
};
template <uint32_t RING>
struct WSrcLdsRing {
    const char *ring; uint32_t base;     // base < RING, row pieces never straddle the end (16-byte granularity)
    template <typename T> __device__ __forceinline__ T ld(uint32_t off) const
    {
        uint32_t a = base + off;
        a = a >= RING ? a - RING : a;
        return *reinterpret_cast<const T *>(ring + a);
    }
};

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
    __device__ __forceinline__ void load(const int8_t *codes, const float *scale, const float *xsum, int lane, int nblk, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = blk0 + lane + 64 * j;
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
    __device__ __forceinline__ void load(const int8_t *codes, const float *scale, const float *, int lane, int nblk, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = blk0 + lane + 64 * j;
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
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(blk0 + lane + 64 * j, nblk - 1);
            c0[j] = nt_load<u32x4>(wrow + (size_t)blk * 32);
            c1[j] = nt_load<u32x4>(wrow + (size_t)blk * 32 + 16);
            sc[j] = nt_load<uint16_t>(wrow + (size_t)nblk * 32 + (size_t)blk * 2);
        }
    }
    template <class S>
    __device__ __forceinline__ void load_src(const S &src, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint32_t blk = (uint32_t)min(blk0 + lane + 64 * j, nblk - 1);
            c0[j] = src.template ld<u32x4>(blk * 32);
            c1[j] = src.template ld<u32x4>(blk * 32 + 16);
            sc[j] = src.template ld<uint16_t>((uint32_t)nblk * 32 + blk * 2);
        }
    }
    __device__ __forceinline__ float dot(const XRegsNat<NJ> &X, float acc0 = 0.0f) const
    {
        float acc = acc0;
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
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(blk0 + lane + 64 * j, nblk - 1);
            c[j][0] = nt_load<u32x4>(wrow + (size_t)blk * 32);
            c[j][1] = nt_load<u32x4>(wrow + (size_t)blk * 32 + 16);
            sb[j] = nt_load<uint32_t>(wrow + (size_t)nblk * 32 + (size_t)blk * 4);
        }
    }
    template <class S>
    __device__ __forceinline__ void load_src(const S &src, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint32_t blk = (uint32_t)min(blk0 + lane + 64 * j, nblk - 1);
            c[j][0] = src.template ld<u32x4>(blk * 32);
            c[j][1] = src.template ld<u32x4>(blk * 32 + 16);
            sb[j] = src.template ld<uint32_t>((uint32_t)nblk * 32 + blk * 4);
        }
    }
    __device__ __forceinline__ float dot(const XRegsB64<NJ> &X, float acc0 = 0.0f) const
    {
        float acc = acc0;
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
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(blk0 + lane + 64 * j, nblk - 1);
            c[j][0] = nt_load<u32x4>(wrow + (size_t)blk * 32);
            c[j][1] = nt_load<u32x4>(wrow + (size_t)blk * 32 + 16);
            hb[j] = nt_load<u32x2>(wrow + (size_t)nblk * 32 + (size_t)blk * 8);
            sb[j] = nt_load<uint32_t>(wrow + (size_t)nblk * 40 + (size_t)blk * 4);
        }
    }
    template <class S>
    __device__ __forceinline__ void load_src(const S &src, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint32_t blk = (uint32_t)min(blk0 + lane + 64 * j, nblk - 1);
            c[j][0] = src.template ld<u32x4>(blk * 32);
            c[j][1] = src.template ld<u32x4>(blk * 32 + 16);
            hb[j] = src.template ld<u32x2>((uint32_t)nblk * 32 + blk * 8);
            sb[j] = src.template ld<uint32_t>((uint32_t)nblk * 40 + blk * 4);
        }
    }
    __device__ __forceinline__ float dot(const XRegsB64<NJ> &X, float acc0 = 0.0f) const
    {
        float acc = acc0;
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
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(blk0 + lane + 64 * j, nblk - 1);
            c[j][0] = nt_load<u32x4>(wrow + (size_t)blk * 32);
            c[j][1] = nt_load<u32x4>(wrow + (size_t)blk * 32 + 16);
            hb[j] = nt_load<u32x4>(wrow + (size_t)nblk * 32 + (size_t)blk * 16);
            sb[j] = nt_load<uint32_t>(wrow + (size_t)nblk * 48 + (size_t)blk * 4);
        }
    }
    template <class S>
    __device__ __forceinline__ void load_src(const S &src, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint32_t blk = (uint32_t)min(blk0 + lane + 64 * j, nblk - 1);
            c[j][0] = src.template ld<u32x4>(blk * 32);
            c[j][1] = src.template ld<u32x4>(blk * 32 + 16);
            hb[j] = src.template ld<u32x4>((uint32_t)nblk * 32 + blk * 16);
            sb[j] = src.template ld<uint32_t>((uint32_t)nblk * 48 + blk * 4);
        }
    }
    // bit pairs at 0,4,8,12 of t -> bits 4-5 of bytes 0..3
    static __device__ __forceinline__ uint32_t spread2(uint32_t t)
    {
        t &= 0x3333u;
        const uint32_t u = (t & 0x0033u) | ((t & 0x3300u) << 8);
        return ((u & 0x00030003u) << 4) | ((u & 0x00300030u) << 8);
    }
    __device__ __forceinline__ float dot(const XRegsB64<NJ> &X, float acc0 = 0.0f) const
    {
        float acc = acc0;
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

// Q3H_B64T1 rows are streamed as nibble pairs (ifa_tiled.h: q3h_aos_to_nibbles), i.e. as WRowQ4B64 with codes 0..10.

// Q3H_NATIVE: the same values at the format's own 32 bytes per block (engine option q3h_native; VERDICT r5 item 5: built to be
// measured against the 36-byte nibble form).  Streamed block = the 28 bytes D0..D6 of q3h_aos_to_tiled (ifa_tiled.h: byte b of
// D[w] = pair code p[4w + b] in its low 7 bits, bit 7 = bit w of p[28 + b]) + (base, scale); row layout
// [D0..D3: 16 B x n][D4, D5: 8 B x n][D6, (base, scale): 8 B x n].  A pair code is p = q0 + 11 q1 (q0 = element 2k, q1 = 2k + 1),
// and   sum q0 xe + q1 xo = sum p xe + sum q1 xo - 11 sum q1 xe   so q0 is never formed; q1 = (93 p + 64) >> 10 exactly for
// p <= 120, two pairs per packed 16-bit multiply-add.  Per dword of four pairs: 8 VALU + 3 dot4 (nibble form: 3 + 2),
// plus 14 VALU per block to gather the four pair codes that live in the top bits.
template <int NJ>
struct WRowQ3HN {
    u32x4 c0[NJ];
    u32x2 c1[NJ], c2[NJ];
    __device__ __forceinline__ void load(const uint8_t *__restrict__ wrow, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int blk = min(blk0 + lane + 64 * j, nblk - 1);
            c0[j] = nt_load<u32x4>(wrow + (size_t)blk * 16);
            c1[j] = nt_load<u32x2>(wrow + (size_t)nblk * 16 + (size_t)blk * 8);
            c2[j] = nt_load<u32x2>(wrow + (size_t)nblk * 24 + (size_t)blk * 8);
        }
    }
    template <class S>
    __device__ __forceinline__ void load_src(const S &src, int nblk, int lane, int blk0 = 0)
    {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint32_t blk = (uint32_t)min(blk0 + lane + 64 * j, nblk - 1);
            c0[j] = src.template ld<u32x4>(blk * 16);
            c1[j] = src.template ld<u32x2>((uint32_t)nblk * 16 + blk * 8);
            c2[j] = src.template ld<u32x2>((uint32_t)nblk * 24 + blk * 8);
        }
    }
    // four pair codes (one per byte, < 121) -> their q1 = p / 11, one per byte: (93 p + 64) >> 10 in packed 16-bit lanes
    // (v_pk_lshrrev_b16 / v_pk_mad_u16: 7 instructions per four pairs)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ uint32_t q1_of(uint32_t P)
    {
        const u16x2 A = __builtin_bit_cast(u16x2, P & 0x00FF00FFu);
        const u16x2 B = __builtin_bit_cast(u16x2, P) >> (unsigned short)8;
        const u16x2 qa = (A * (unsigned short)93 + (unsigned short)64) >> (unsigned short)10;
        const u16x2 qb = (B * (unsigned short)93 + (unsigned short)64) >> (unsigned short)10;
        return __builtin_bit_cast(uint32_t, qa) | (__builtin_bit_cast(uint32_t, qb) << 8);
    }
    static __device__ __forceinline__ void pair4(uint32_t P, int xe, int xo, int &dp, int &dq)
    {
        const uint32_t Q = q1_of(P);
        dp = sdot4((int)P, xe, dp);
        dp = sdot4((int)Q, xo, dp);
        dq = sdot4((int)Q, xe, dq);
    }
    __device__ __forceinline__ float dot(const XRegsB64<NJ> &X, float acc0 = 0.0f) const
    {
        float acc = acc0;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            // (scalar temporaries: a local array initialised from vector registers goes to scratch memory on this compiler)
            const uint32_t D0 = c0[j][0], D1 = c0[j][1], D2 = c0[j][2], D3 = c0[j][3], D4 = c1[j][0], D5 = c1[j][1], D6 = c2[j][0];
            const uint32_t sbw = c2[j][1];
            const float base = hbits2f((uint16_t)(sbw & 0xFFFFu)), scale = hbits2f((uint16_t)(sbw >> 16));
            const uint32_t E = ((D0 >> 7) & 0x01010101u) | ((D1 >> 6) & 0x02020202u) | ((D2 >> 5) & 0x04040404u) | ((D3 >> 4) & 0x08080808u)
                             | ((D4 >> 3) & 0x10101010u) | ((D5 >> 2) & 0x20202020u) | ((D6 >> 1) & 0x40404040u);
            const uint32_t m7 = 0x7F7F7F7Fu;
            int dp0 = 0, dq0 = 0, dp1 = 0, dq1 = 0;
            pair4(D0 & m7, X.xe[j][0][0], X.xo[j][0][0], dp0, dq0);
            pair4(D1 & m7, X.xe[j][0][1], X.xo[j][0][1], dp0, dq0);
            pair4(D2 & m7, X.xe[j][0][2], X.xo[j][0][2], dp0, dq0);
            pair4(D3 & m7, X.xe[j][0][3], X.xo[j][0][3], dp0, dq0);
            pair4(D4 & m7, X.xe[j][1][0], X.xo[j][1][0], dp1, dq1);
            pair4(D5 & m7, X.xe[j][1][1], X.xo[j][1][1], dp1, dq1);
            pair4(D6 & m7, X.xe[j][1][2], X.xo[j][1][2], dp1, dq1);
            pair4(E, X.xe[j][1][3], X.xo[j][1][3], dp1, dq1);
            acc = acc + dec_term(dp0 - 11 * dq0, scale, base, X.xsf[j][0], X.xs[j][0]);
            acc = acc + dec_term(dp1 - 11 * dq1, scale, base, X.xsf[j][1], X.xs[j][1]);
        }
        return acc;
    }
};

} // namespace ifa
