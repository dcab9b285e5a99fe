/*
 * ifa_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's quantized decode path: block codecs,
 * activation quantizer, quant x int8 / quant x fp16 GEMV arithmetic, norms,
 * RoPE/ALiBi, masked softmax, activations, KV-cache attention.
 * See ifa_oracle.h for the rules on who may call this.  Whole-model decode is
 * in ifa_oracle_model.c.
 *
 * Arithmetic notes
 *  - All fp32 expressions are written so that no FMA contraction changes them
 *    (build with -ffp-contract=off); this matches the reference header as
 *    compiled for the host by g++ (the pin in tests/golden).  What nvcc's
 *    default -fmad=true would do on a CUDA device is not reproducible here and
 *    is stated as "unpinned" in DESIGN.md.
 *  - fp16 <-> fp32 follows IEEE round-to-nearest-even, like CUDA __float2half_rn
 *    and half_float 2.2.0 (3rd_party/half/half.hpp:373-374).
 */
#include "ifa_oracle.h"
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <string.h>
#include <stdlib.h>

/* ------------------------------------------------------------------ fp16 */
float orc_h2f(orc_f16 h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

orc_f16 orc_f2h(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) { /* inf / nan */
        if (ax > 0x7F800000u) return (orc_f16)(sign | 0x7E00u | ((ax >> 13) & 0x3FFu));
        return (orc_f16)(sign | 0x7C00u);
    }
    if (ax >= 0x477FF000u) { /* >= 65520 rounds to inf */
        return (orc_f16)(sign | 0x7C00u);
    }
    if (ax < 0x38800000u) { /* subnormal half or zero */
        if (ax < 0x33000000u) { /* < 2^-25 -> 0 (ties-to-even at exactly 2^-25 -> 0) */
            return (orc_f16)sign;
        }
        uint32_t e = ax >> 23;                 /* biased exponent */
        uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
        uint32_t shift = 126 - e;              /* 14..24 */
        /* value = m * 2^(e-150); half subnormal unit = 2^-24 -> q = m >> (shift) */
        uint32_t q = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) q++;
        return (orc_f16)(sign | q);
    }
    {
        uint32_t e = (ax >> 23) - 127 + 15;
        uint32_t m = ax & 0x7FFFFFu;
        uint32_t q = (e << 10) | (m >> 13);
        uint32_t rem = m & 0x1FFFu;
        if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
        return (orc_f16)(sign | q);
    }
}

void orc_h2f_n(const orc_f16 *src, float *dst, size_t n)
{
    for (size_t i = 0; i < n; i++) dst[i] = orc_h2f(src[i]);
}
void orc_f2h_n(const float *src, orc_f16 *dst, size_t n)
{
    for (size_t i = 0; i < n; i++) dst[i] = orc_f2h(src[i]);
}

/* ------------------------------------------------------- format registry */
/* src/common/quant_types.h:11-174; sizes: SURVEY.md facts (sizeof checks).   */
int orc_block_capacity(int dtype)
{
    switch (dtype) {
    case ORC_F32: case ORC_F16: return 1;
    case ORC_Q8_B32T1: case ORC_Q8_B32T2: case ORC_Q5_B32T1: case ORC_Q4_B32T1A:
    case ORC_Q4_B32T1B: case ORC_Q3_B32T1A: case ORC_Q3_B32T1B: case ORC_Q2_B32T1A:
    case ORC_Q2_B32T1B: return 32;
    case ORC_Q4_B16: return 16;
    case ORC_Q6_B64T1: case ORC_Q5_B64T1: case ORC_Q4_B64T1: case ORC_Q3H_B64T1: return 64;
    default: return 0;
    }
}

int orc_block_bytes(int dtype)
{
    switch (dtype) {
    case ORC_F32: return 4;
    case ORC_F16: return 2;
    case ORC_Q8_B32T1: return 36;
    case ORC_Q8_B32T2: return 34;
    case ORC_Q6_B64T1: return 52;
    case ORC_Q5_B64T1: return 44;
    case ORC_Q5_B32T1: return 24;
    case ORC_Q4_B16: return 10;
    case ORC_Q4_B32T1A: case ORC_Q4_B32T1B: return 20;
    case ORC_Q4_B64T1: return 36;
    case ORC_Q3H_B64T1: return 32;
    case ORC_Q3_B32T1A: case ORC_Q3_B32T1B: return 16;
    case ORC_Q2_B32T1A: case ORC_Q2_B32T1B: return 12;
    default: return 0;
    }
}

size_t orc_row_bytes(int dtype, size_t cols)
{
    int cap = orc_block_capacity(dtype);
    if (cap <= 0) return 0;
    return (cols + (size_t)cap - 1) / (size_t)cap * (size_t)orc_block_bytes(dtype);
}

static inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline void wr16(uint8_t *p, uint16_t v) { p[0] = (uint8_t)(v & 0xFF); p[1] = (uint8_t)(v >> 8); }

static void value_range(const float *a, int n, float *mn, float *mx)
{   /* Quantization::GetValueRange, quantization.h:68-85 */
    *mn = a[0]; *mx = a[0];
    for (int i = 1; i < n; i++) {
        float v = a[i];
        if (*mn > v) *mn = v;
        if (*mx < v) *mx = v;
    }
}

static inline uint32_t qmin_u32(uint32_t a, uint32_t b) { return a <= b ? a : b; }

/* ------------------------------------------------------ block quantizers */
/* Each takes one block of fp32 source values (already widened from the
 * source type, exactly as the reference's (float)source[i]).             */

/* quantization.h:110-152 */
static void q_block_q8_b32t1(const float *s, uint8_t *b)
{
    float mn, mx; value_range(s, 32, &mn, &mx);
    float scale = (mx - mn) / 255;
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    wr16(b, orc_f2h(mn)); wr16(b + 2, orc_f2h(scale));
    for (int r = 0; r < 32; r++) {
        float qf = (s[r] - mn) * inv;
        uint32_t q = (uint32_t)(qf + 0.5f);
        q = q > 255 ? 255 : q;
        b[4 + r] = (uint8_t)q;
    }
}

/* quantization.h:186-221 (host-style; see orc_quantize_act_q8 for the device one) */
static void q_block_q8_b32t2_host(const float *s, uint8_t *b)
{
    float mn, mx; value_range(s, 32, &mn, &mx);
    float m1 = fabsf(mx), m2 = fabsf(mn);
    float m0 = m1 > m2 ? m1 : m2;
    float scale = m0 / 127;
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    wr16(b, orc_f2h(scale));
    for (int r = 0; r < 32; r++) {
        float qf = s[r] * inv;
        int q = (int)round((double)qf);
        q = q > 127 ? 127 : q;
        q = q < -128 ? -128 : q;
        b[2 + r] = (uint8_t)(int8_t)q;
    }
}

/* Tensor_QuantizeQ8_B32T2_Alg2_Kernel, src/kernels/tensor_quant.h:44-82 */
static void q_block_q8_b32t2_dev(const float *s, int n_valid, uint8_t *b)
{
    float mxv = 0.0f;
    for (int r = 0; r < 32; r++) {
        float v = r < n_valid ? fabsf(s[r]) : 0.0f;
        mxv = mxv > v ? mxv : v;
    }
    float scale = mxv / 127;
    for (int r = 0; r < n_valid; r++) {
        int q = scale <= 0.000001f ? 0 : (int)roundf(s[r] / scale);
        q = q < -128 ? -128 : q;
        q = q > 127 ? 127 : q;
        b[2 + r] = (uint8_t)(int8_t)q;
    }
    wr16(b, orc_f2h(scale));
}

/* quantization.h:268-322 (note: divides by quant_num-1 = 62, appendix A1) */
static void q_block_q6_b64t1(const float *s, uint8_t *b)
{
    float mn, mx; value_range(s, 64, &mn, &mx);
    const int quant_num = 63; const uint32_t max_q = 63;
    float scale = (mx - mn) / (quant_num - 1);
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    wr16(b, orc_f2h(mn)); wr16(b + 2, orc_f2h(scale));
    uint8_t *dh = b + 4, *d = b + 4 + 16;
    for (int r = 0; r < 16; r++) {
        uint32_t q[4];
        for (int i = 0; i < 4; i++) {
            float v = (s[4 * r + i] - mn) * inv;
            q[i] = qmin_u32((uint32_t)(v + 0.5f), max_q);
        }
        uint32_t qh = (q[0] >> 4) | ((q[1] & 0x30) >> 2) | (q[2] & 0x30) | ((q[3] & 0x30) << 2);
        dh[r] = (uint8_t)qh;
        d[2 * r] = (uint8_t)((q[0] & 0x0F) | ((q[1] & 0x0F) << 4));
        d[2 * r + 1] = (uint8_t)((q[2] & 0x0F) | ((q[3] & 0x0F) << 4));
    }
}

/* Quantization::QuantizeQ5Row, quantization.h:348-393 */
static void q_block_q5_b32t1(const float *s, uint8_t *b)
{
    float mn, mx; value_range(s, 32, &mn, &mx);
    float delta = (mx - mn) / 31;
    float inv = delta >= 0.00001f ? (1.0f / delta) : 0.0f;
    /* struct order: scale[2], base[2], h_data[4], data[16] */
    wr16(b, orc_f2h(delta)); wr16(b + 2, orc_f2h(mn));
    uint32_t qh = 0;
    for (int r = 0; r < 16; r++) {
        float v1 = (s[r] - mn) * inv;
        float v2 = (s[r + 16] - mn) * inv;
        uint32_t q1 = (uint32_t)(v1 + 0.5f);
        uint32_t q2 = (uint32_t)(v2 + 0.5f);
        b[8 + r] = (uint8_t)((q1 & 0x0F) | ((q2 & 0x0F) << 4));
        qh |= (((q1 & 0x10) >> 4) << r);
        qh |= (((q2 & 0x10) >> 4) << (r + 16));
    }
    b[4] = (uint8_t)(qh & 0xFF); b[5] = (uint8_t)((qh >> 8) & 0xFF);
    b[6] = (uint8_t)((qh >> 16) & 0xFF); b[7] = (uint8_t)((qh >> 24) & 0xFF);
}

/* quantization.h:446-503 (divides by quant_num-1 = 30) */
static void q_block_q5_b64t1(const float *s, uint8_t *b)
{
    float mn, mx; value_range(s, 64, &mn, &mx);
    const int quant_num = 31; const uint32_t max_q = 31;
    float scale = (mx - mn) / (quant_num - 1);
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    wr16(b, orc_f2h(mn)); wr16(b + 2, orc_f2h(scale));
    uint8_t *dh = b + 4, *d = b + 4 + 8;
    for (int r = 0; r < 8; r++) {
        uint32_t q[8];
        for (int li = 0; li < 8; li++) {
            float v = (s[8 * r + li] - mn) * inv;
            q[li] = qmin_u32((uint32_t)(v + 0.5f), max_q);
        }
        uint32_t qh = ((q[0] & 0x10) >> 4) | ((q[1] & 0x10) >> 3) | ((q[2] & 0x10) >> 2)
            | ((q[3] & 0x10) >> 1) | ((q[4] & 0x10)) | ((q[5] & 0x10) << 1)
            | ((q[6] & 0x10) << 2) | ((q[7] & 0x10) << 3);
        dh[r] = (uint8_t)qh;
        d[4 * r] = (uint8_t)((q[0] & 0x0F) | ((q[1] & 0x0F) << 4));
        d[4 * r + 1] = (uint8_t)((q[2] & 0x0F) | ((q[3] & 0x0F) << 4));
        d[4 * r + 2] = (uint8_t)((q[4] & 0x0F) | ((q[5] & 0x0F) << 4));
        d[4 * r + 3] = (uint8_t)((q[6] & 0x0F) | ((q[7] & 0x0F) << 4));
    }
}

/* quantization.h:535-586 (A) and :589-632 (B) */
static void q_block_q4_b32t1(const float *s, uint8_t *b, int variant_b)
{
    float mn, mx; value_range(s, 32, &mn, &mx);
    float scale, base, round_add;
    if (!variant_b) { scale = (mx - mn) / 15; base = mn; round_add = 0.5f; }
    else { scale = (mx - mn) / 16; base = 0; round_add = 0.0001f; }
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    if (variant_b) base = mn + 0.5f * scale;
    wr16(b, orc_f2h(base)); wr16(b + 2, orc_f2h(scale));
    for (int r = 0; r < 16; r++) {
        float v1 = (s[2 * r] - mn) * inv;
        float v2 = (s[2 * r + 1] - mn) * inv;
        uint32_t q1 = (uint32_t)(v1 + round_add);
        uint32_t q2 = (uint32_t)(v2 + round_add);
        q1 = q1 > 15 ? 15 : q1;
        q2 = q2 > 15 ? 15 : q2;
        b[4 + r] = (uint8_t)((q1 & 0x0F) | ((q2 & 0x0F) << 4));
    }
}

/* quantization.h:41-66 helpers + :657-712 */
static uint8_t enc_scale_u8(float scale) { return (uint8_t)(scale * 1000 + 0.5f); }
static float dec_scale_u8(uint8_t u) { return (float)u / 1000; }
static uint8_t enc_base_u8(float base) { return (uint8_t)(base * 100 + 100.5f); }
static float dec_base_u8(uint8_t u) { return (int)(uint32_t)u / 100.0f - 1.0f; }
static float adjust_base(float base) { uint8_t u8 = (uint8_t)(base * 100 + 100.01); return dec_base_u8(u8); }

static void q_block_q4_b16(const float *s, uint8_t *b)
{
    float mn, mx; value_range(s, 16, &mn, &mx);
    mn = adjust_base(mn);
    float scale = (mx - mn) / 15;
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    b[0] = enc_base_u8(mn);
    b[1] = enc_scale_u8(scale);
    for (int r = 0; r < 8; r++) {
        float v1 = (s[2 * r] - mn) * inv;
        float v2 = (s[2 * r + 1] - mn) * inv;
        uint32_t q1 = (uint32_t)(v1 + 0.5f);
        uint32_t q2 = (uint32_t)(v2 + 0.5f);
        q1 = q1 > 15 ? 15 : q1;
        q2 = q2 > 15 ? 15 : q2;
        b[2 + r] = (uint8_t)((q1 & 0x0F) | ((q2 & 0x0F) << 4));
    }
}

/* quantization.h:757-803 (divides by quant_num-1 = 14) */
static void q_block_q4_b64t1(const float *s, uint8_t *b)
{
    float mn, mx; value_range(s, 64, &mn, &mx);
    const int quant_num = 15; const uint32_t max_q = 15;
    float scale = (mx - mn) / (quant_num - 1);
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    wr16(b, orc_f2h(mn)); wr16(b + 2, orc_f2h(scale));
    for (int r = 0; r < 16; r++) {
        uint32_t q[4];
        for (int i = 0; i < 4; i++) {
            float v = (s[4 * r + i] - mn) * inv;
            q[i] = qmin_u32((uint32_t)(v + 0.5f), max_q);
        }
        b[4 + 2 * r] = (uint8_t)(q[0] | (q[1] << 4));
        b[4 + 2 * r + 1] = (uint8_t)(q[2] | (q[3] << 4));
    }
}

/* quantization.h:854-926: 11 levels, pairs packed base-11 into 7 bits */
static void q_block_q3h_b64t1(const float *s, uint8_t *b)
{
    float mn, mx; value_range(s, 64, &mn, &mx);
    const int quant_num = 11;
    float scale = (mx - mn) / (quant_num - 1);
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    wr16(b, orc_f2h(mn)); wr16(b + 2, orc_f2h(scale));
    uint8_t *dh = b + 4, *dm = b + 8, *d = b + 16;
    int data_h = 0;
    for (int r = 0; r < 8; r++) {
        int qa[8];
        for (int i = 0; i < 8; i++) {
            float v = (s[8 * r + i] - mn) * inv;
            qa[i] = (int)(v + 0.5f);
            if (qa[i] < 0) qa[i] = 0;
            if (qa[i] > quant_num - 1) qa[i] = quant_num - 1;
        }
        int q1 = qa[0] + qa[1] * 11, q2 = qa[2] + qa[3] * 11;
        int q3 = qa[4] + qa[5] * 11, q4 = qa[6] + qa[7] * 11;
        d[2 * r] = (uint8_t)((q1 & 0x0F) | ((q2 & 0x0F) << 4));
        d[2 * r + 1] = (uint8_t)((q3 & 0x0F) | ((q4 & 0x0F) << 4));
        dm[r] = (uint8_t)(((q1 & 0x30) >> 4) | ((q2 & 0x30) >> 2) | (q3 & 0x30) | ((q4 & 0x30) << 2));
        if (r % 2 == 0) {
            data_h = ((q1 & 0x40) >> 6) | ((q2 & 0x40) >> 5) | ((q3 & 0x40) >> 4) | ((q4 & 0x40) >> 3);
        } else {
            data_h = data_h | ((q1 & 0x40) >> 2) | ((q2 & 0x40) >> 1) | (q3 & 0x40) | ((q4 & 0x40) << 1);
            dh[r / 2] = (uint8_t)data_h;
            data_h = 0;
        }
    }
}

/* quantization.h:964-1014 (A), :1017-1068 (B) */
static void q_block_q3_b32t1(const float *s, uint8_t *b, int variant_b)
{
    float mn, mx; value_range(s, 32, &mn, &mx);
    float scale, base, round_add;
    if (!variant_b) { scale = (mx - mn) / 7; base = mn; round_add = 0.5f; }
    else { scale = (mx - mn) / 8; base = 0; round_add = 0.0001f; }
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    if (variant_b) base = mn + 0.5f * scale;
    wr16(b, orc_f2h(base)); wr16(b + 2, orc_f2h(scale));
    uint8_t *hd = b + 4, *d = b + 8;
    for (int r = 0; r < 4; r++) {
        uint32_t q[8];
        for (int i = 0; i < 8; i++) {
            float v = (s[8 * r + i] - mn) * inv;
            q[i] = (uint32_t)(v + round_add);
            if (q[i] > 7) q[i] = 7;
        }
        d[2 * r] = (uint8_t)((q[0] & 3) | ((q[1] & 3) << 2) | ((q[2] & 3) << 4) | ((q[3] & 3) << 6));
        d[2 * r + 1] = (uint8_t)((q[4] & 3) | ((q[5] & 3) << 2) | ((q[6] & 3) << 4) | ((q[7] & 3) << 6));
        hd[r] = (uint8_t)(((q[0] & 4) >> 2) | ((q[1] & 4) >> 1) | (q[2] & 4) | ((q[3] & 4) << 1)
            | ((q[4] & 4) << 2) | ((q[5] & 4) << 3) | ((q[6] & 4) << 4) | ((q[7] & 4) << 5));
    }
}

/* quantization.h:1098-1144 (A), :1147-1193 (B) */
static void q_block_q2_b32t1(const float *s, uint8_t *b, int variant_b)
{
    float mn, mx; value_range(s, 32, &mn, &mx);
    float scale, base, round_add;
    if (!variant_b) { scale = (mx - mn) / 3; base = mn; round_add = 0.5f; }
    else { scale = (mx - mn) / 4; base = 0; round_add = 0.0001f; }
    float inv = scale >= 0.00001f ? (1.0f / scale) : 0.0f;
    if (variant_b) base = mn + 0.5f * scale;
    wr16(b, orc_f2h(base)); wr16(b + 2, orc_f2h(scale));
    for (int r = 0; r < 8; r++) {
        uint32_t q[4];
        for (int i = 0; i < 4; i++) {
            float v = (s[4 * r + i] - mn) * inv;
            q[i] = (uint32_t)(v + round_add);
            q[i] = q[i] > 3 ? 3 : q[i];
        }
        b[4 + r] = (uint8_t)(q[0] | (q[1] << 2) | (q[2] << 4) | (q[3] << 6));
    }
}

static int quantize_block(int dtype, const float *s, uint8_t *b)
{
    switch (dtype) {
    case ORC_Q8_B32T1: q_block_q8_b32t1(s, b); return 0;
    /* TensorOpr::Quantize always picks alg 2 for Q8_B32T2 (tensor_opr.cu:1647-1648), appendix A3 */
    case ORC_Q8_B32T2: q_block_q8_b32t2_dev(s, 32, b); return 0;
    case ORC_Q6_B64T1: q_block_q6_b64t1(s, b); return 0;
    case ORC_Q5_B64T1: q_block_q5_b64t1(s, b); return 0;
    case ORC_Q5_B32T1: q_block_q5_b32t1(s, b); return 0;
    case ORC_Q4_B16: q_block_q4_b16(s, b); return 0;
    case ORC_Q4_B32T1A: q_block_q4_b32t1(s, b, 0); return 0;
    case ORC_Q4_B32T1B: q_block_q4_b32t1(s, b, 1); return 0;
    case ORC_Q4_B64T1: q_block_q4_b64t1(s, b); return 0;
    case ORC_Q3H_B64T1: q_block_q3h_b64t1(s, b); return 0;
    case ORC_Q3_B32T1A: q_block_q3_b32t1(s, b, 0); return 0;
    case ORC_Q3_B32T1B: q_block_q3_b32t1(s, b, 1); return 0;
    case ORC_Q2_B32T1A: q_block_q2_b32t1(s, b, 0); return 0;
    case ORC_Q2_B32T1B: q_block_q2_b32t1(s, b, 1); return 0;
    default: return -1;
    }
}

int orc_quantize_rows_f32(int dtype, const float *src, size_t rows, size_t cols, uint8_t *dst)
{
    int cap = orc_block_capacity(dtype), bb = orc_block_bytes(dtype);
    if (cap <= 1 || cols % (size_t)cap != 0) return -1;
    size_t nb = cols / (size_t)cap;
    for (size_t r = 0; r < rows; r++) {
        for (size_t k = 0; k < nb; k++) {
            if (quantize_block(dtype, src + r * cols + k * (size_t)cap,
                               dst + (r * nb + k) * (size_t)bb) != 0) return -1;
        }
    }
    return 0;
}

int orc_quantize_rows(int dtype, const orc_f16 *src, size_t rows, size_t cols, uint8_t *dst)
{
    int cap = orc_block_capacity(dtype), bb = orc_block_bytes(dtype);
    if (cap <= 1 || cols % (size_t)cap != 0) return -1;
    size_t nb = cols / (size_t)cap;
    float tmp[64];
    for (size_t r = 0; r < rows; r++) {
        for (size_t k = 0; k < nb; k++) {
            orc_h2f_n(src + r * cols + k * (size_t)cap, tmp, (size_t)cap);
            if (quantize_block(dtype, tmp, dst + (r * nb + k) * (size_t)bb) != 0) return -1;
        }
    }
    return 0;
}

int orc_quantize_act_q8(const orc_f16 *src, size_t rows, size_t cols, uint8_t *dst)
{
    size_t nb = (cols + 31) / 32;
    float tmp[32];
    for (size_t r = 0; r < rows; r++) {
        for (size_t k = 0; k < nb; k++) {
            int nv = (int)(cols - k * 32 < 32 ? cols - k * 32 : 32);
            for (int i = 0; i < 32; i++) tmp[i] = i < nv ? orc_h2f(src[r * cols + k * 32 + (size_t)i]) : 0.0f;
            q_block_q8_b32t2_dev(tmp, nv, dst + (r * nb + k) * 34);
        }
    }
    return 0;
}

int orc_quantize_q8_b32t2_host(const orc_f16 *src, size_t rows, size_t cols, uint8_t *dst)
{
    if (cols % 32 != 0) return -1;
    size_t nb = cols / 32;
    float tmp[32];
    for (size_t r = 0; r < rows; r++)
        for (size_t k = 0; k < nb; k++) {
            orc_h2f_n(src + r * cols + k * 32, tmp, 32);
            q_block_q8_b32t2_host(tmp, dst + (r * nb + k) * 34);
        }
    return 0;
}

/* ---------------------------------------------- block decode (codes) */
/* Returns integer codes in element order plus fp32 scale/base, i.e. the
 * value of element i is codes[i]*scale + base.  Restates the Dequantize*
 * and GetInt4 members of quantization.h (lines cited per case).         */
static int decode_block(int dtype, const uint8_t *b, int32_t *q, float *scale, float *base)
{
    switch (dtype) {
    case ORC_Q8_B32T1: /* quantization.h:94-108 */
        *base = orc_h2f(rd16(b)); *scale = orc_h2f(rd16(b + 2));
        for (int i = 0; i < 32; i++) q[i] = b[4 + i];
        return 32;
    case ORC_Q8_B32T2: /* :171-184, GetInt4 :160-168 */
        *scale = orc_h2f(rd16(b)); *base = 0.0f;
        for (int i = 0; i < 32; i++) q[i] = (int8_t)b[2 + i];
        return 32;
    case ORC_Q6_B64T1: { /* :240-266, GetInt4 :227-237 */
        *base = orc_h2f(rd16(b)); *scale = orc_h2f(rd16(b + 2));
        const uint8_t *dh = b + 4, *d = b + 20;
        for (int idx = 0; idx < 16; idx++) {
            uint8_t qh = dh[idx];
            uint16_t qd = rd16(d + 2 * idx);
            q[4 * idx] = ((qd) & 0x0F) | ((qh & 0x03) << 4);
            q[4 * idx + 1] = ((qd >> 4) & 0x0F) | (((qh >> 2) & 0x03) << 4);
            q[4 * idx + 2] = ((qd >> 8) & 0x0F) | (((qh >> 4) & 0x03) << 4);
            q[4 * idx + 3] = ((qd >> 12) & 0x0F) | (((qh >> 6) & 0x03) << 4);
        }
        return 64; }
    case ORC_Q5_B64T1: { /* :414-443, GetInt4 :401-411 */
        *base = orc_h2f(rd16(b)); *scale = orc_h2f(rd16(b + 2));
        const uint8_t *dh = b + 4, *d = b + 12;
        for (int idx = 0; idx < 16; idx++) {
            uint8_t qh = dh[idx / 2];
            if (idx % 2 != 0) qh = (uint8_t)(qh >> 4);
            uint16_t qd = rd16(d + 2 * idx);
            q[4 * idx] = ((qd) & 0x0F) | ((qh & 0x01) << 4);
            q[4 * idx + 1] = ((qd >> 4) & 0x0F) | (((qh >> 1) & 0x01) << 4);
            q[4 * idx + 2] = ((qd >> 8) & 0x0F) | (((qh >> 2) & 0x01) << 4);
            q[4 * idx + 3] = ((qd >> 12) & 0x0F) | (((qh >> 3) & 0x01) << 4);
        }
        return 64; }
    case ORC_Q5_B32T1: { /* :325-345 */
        *scale = orc_h2f(rd16(b)); *base = orc_h2f(rd16(b + 2));
        uint32_t qh = (uint32_t)b[4] | ((uint32_t)b[5] << 8) | ((uint32_t)b[6] << 16) | ((uint32_t)b[7] << 24);
        for (int idx = 0; idx < 16; idx++) {
            uint8_t xh0 = (qh >> idx) & 1, xh1 = (qh >> (idx + 16)) & 1;
            q[idx] = (b[8 + idx] & 0x0F) | (xh0 << 4);
            q[idx + 16] = (b[8 + idx] >> 4) | (xh1 << 4);
        }
        return 32; }
    case ORC_Q4_B16: /* :638-655 */
        *base = dec_base_u8(b[0]); *scale = dec_scale_u8(b[1]);
        for (int i = 0; i < 8; i++) { q[2 * i] = b[2 + i] & 0x0F; q[2 * i + 1] = b[2 + i] >> 4; }
        return 16;
    case ORC_Q4_B32T1A: case ORC_Q4_B32T1B: /* :516-533, GetInt4 :506-513 */
        *base = orc_h2f(rd16(b)); *scale = orc_h2f(rd16(b + 2));
        for (int i = 0; i < 16; i++) { q[2 * i] = b[4 + i] & 0x0F; q[2 * i + 1] = b[4 + i] >> 4; }
        return 32;
    case ORC_Q4_B64T1: /* :735-754, GetInt4 :714-721 */
        *base = orc_h2f(rd16(b)); *scale = orc_h2f(rd16(b + 2));
        for (int i = 0; i < 32; i++) { q[2 * i] = b[4 + i] & 0x0F; q[2 * i + 1] = b[4 + i] >> 4; }
        return 64;
    case ORC_Q3H_B64T1: { /* :823-851, GetInt4 :809-820 */
        *base = orc_h2f(rd16(b)); *scale = orc_h2f(rd16(b + 2));
        const uint8_t *dh = b + 4, *dm = b + 8, *d = b + 16;
        for (int idx = 0; idx < 8; idx++) {
            uint16_t u16 = rd16(d + 2 * idx);
            uint8_t m8 = dm[idx];
            uint8_t h8 = idx % 2 == 0 ? (dh[idx / 2] & 0x0F) : ((dh[idx / 2] & 0xF0) >> 4);
            int q0 = ((u16 & 0x000F)) | ((m8 & 0x03) << 4) | ((h8 & 0x01) << 6);
            int q1 = ((u16 & 0x00F0) >> 4) | ((m8 & 0x0C) << 2) | ((h8 & 0x02) << 5);
            int q2 = ((u16 & 0x0F00) >> 8) | ((m8 & 0x30)) | ((h8 & 0x04) << 4);
            int q3 = ((u16 & 0xF000) >> 12) | ((m8 & 0xC0) >> 2) | ((h8 & 0x08) << 3);
            q[8 * idx] = q0 % 11; q[8 * idx + 1] = q0 / 11;
            q[8 * idx + 2] = q1 % 11; q[8 * idx + 3] = q1 / 11;
            q[8 * idx + 4] = q2 % 11; q[8 * idx + 5] = q2 / 11;
            q[8 * idx + 6] = q3 % 11; q[8 * idx + 7] = q3 / 11;
        }
        return 64; }
    case ORC_Q3_B32T1A: case ORC_Q3_B32T1B: { /* :933-961 */
        *base = orc_h2f(rd16(b)); *scale = orc_h2f(rd16(b + 2));
        const uint8_t *hd = b + 4, *d = b + 8;
        for (int idx = 0; idx < 4; idx++) {
            uint16_t u16 = rd16(d + 2 * idx);
            uint8_t h8 = hd[idx];
            for (int i = 0; i < 8; i++)
                q[8 * idx + i] = ((u16 >> (2 * i)) & 3) | (((h8 >> i) & 1) << 2);
        }
        return 32; }
    case ORC_Q2_B32T1A: case ORC_Q2_B32T1B: /* :1075-1095 */
        *base = orc_h2f(rd16(b)); *scale = orc_h2f(rd16(b + 2));
        for (int i = 0; i < 8; i++) {
            q[4 * i] = b[4 + i] & 3; q[4 * i + 1] = (b[4 + i] >> 2) & 3;
            q[4 * i + 2] = (b[4 + i] >> 4) & 3; q[4 * i + 3] = b[4 + i] >> 6;
        }
        return 32;
    default:
        return -1;
    }
}

int orc_unpack_codes(int dtype, const uint8_t *src, size_t rows, size_t cols, int32_t *codes)
{
    int cap = orc_block_capacity(dtype), bb = orc_block_bytes(dtype);
    if (cap <= 1 || cols % (size_t)cap != 0) return -1;
    size_t nb = cols / (size_t)cap;
    float s, bs;
    for (size_t r = 0; r < rows; r++)
        for (size_t k = 0; k < nb; k++)
            if (decode_block(dtype, src + (r * nb + k) * (size_t)bb, codes + r * cols + k * (size_t)cap, &s, &bs) < 0)
                return -1;
    return 0;
}

int orc_dequantize_rows_f32(int dtype, const uint8_t *src, size_t rows, size_t cols, float *dst)
{
    int cap = orc_block_capacity(dtype), bb = orc_block_bytes(dtype);
    if (cap <= 1 || cols % (size_t)cap != 0) return -1;
    size_t nb = cols / (size_t)cap;
    int32_t q[64]; float s, bs;
    for (size_t r = 0; r < rows; r++)
        for (size_t k = 0; k < nb; k++) {
            if (decode_block(dtype, src + (r * nb + k) * (size_t)bb, q, &s, &bs) < 0) return -1;
            float *o = dst + r * cols + k * (size_t)cap;
            if (dtype == ORC_Q8_B32T2) {
                for (int i = 0; i < cap; i++) o[i] = (float)q[i] * s;           /* q * scale */
            } else {
                for (int i = 0; i < cap; i++) { float t = (float)q[i] * s; o[i] = t + bs; }
            }
        }
    return 0;
}

int orc_dequantize_rows(int dtype, const uint8_t *src, size_t rows, size_t cols, orc_f16 *dst)
{
    float *tmp = (float *)malloc(sizeof(float) * cols);
    if (!tmp) return -1;
    size_t rb = orc_row_bytes(dtype, cols);
    for (size_t r = 0; r < rows; r++) {
        if (orc_dequantize_rows_f32(dtype, src + r * rb, 1, cols, tmp) != 0) { free(tmp); return -1; }
        orc_f2h_n(tmp, dst + r * cols, cols);
    }
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------ GEMV */
/* int8 x intN path: src/kernels/gemv.h:1499-1709, launch shapes
 * src/tensor/tensor_mul.cu:1097-1221.  A 32-lane warp walks 8 weight blocks
 * per iteration (4 lanes per block); each lane adds scale*isum*xs then
 * base*xsum*xs in fp32; xor-butterfly over 32 lanes; half out.            */
static int ax8_eligible(int dtype)
{
    switch (dtype) {
    case ORC_Q8_B32T2: case ORC_Q6_B64T1: case ORC_Q5_B64T1: case ORC_Q4_B32T1A:
    case ORC_Q4_B32T1B: case ORC_Q4_B64T1: case ORC_Q3H_B64T1: return 1;
    default: return 0;
    }
}

/* default OpenMP team size of this library (oracle.py passes the CPUs the process may really use) */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* tests switch the specialised loops off to compare them with the general ones */
static int g_orc_slow_paths = 0;
void orc_set_slow_paths(int on) { g_orc_slow_paths = on; }

int orc_gemv_ax8(int dtype_w, const uint8_t *W, size_t rows, size_t cols,
                 const uint8_t *xq8, orc_f16 *y, double *y_f64)
{
    if (!ax8_eligible(dtype_w)) return -1;
    int cap = orc_block_capacity(dtype_w), bb = orc_block_bytes(dtype_w);
    if (cols % (size_t)cap != 0) return -1;
    size_t nb = cols / (size_t)cap;
    size_t nxb = cols / 32;
    /* decode x once */
    int32_t *xq = (int32_t *)malloc(sizeof(int32_t) * cols);
    float *xs = (float *)malloc(sizeof(float) * nxb);
    if (!xq || !xs) { free(xq); free(xs); return -1; }
    for (size_t k = 0; k < nxb; k++) {
        float s, b0;
        decode_block(ORC_Q8_B32T2, xq8 + k * 34, xq + k * 32, &s, &b0);
        xs[k] = s;
    }
    const int per_lane = cap == 32 ? 8 : 16; /* elements a lane covers per block */
    /* Q4_B32T1 without the fp64 shadow sum: the same float operations in the same order as the general loop below
     * (tests/test_oracle_cpu.py checks the two bit for bit), with the nibbles multiplied straight out of the block and the
     * per-part sums of x hoisted out of the row loop -- this is the loop bench.py's cpu_baseline spends its time in */
    if ((dtype_w == ORC_Q4_B32T1A || dtype_w == ORC_Q4_B32T1B) && !y_f64 && !g_orc_slow_paths) {
        int32_t *xsum = (int32_t *)malloc(sizeof(int32_t) * nb * 4);
        if (!xsum) { free(xq); free(xs); return -1; }
        for (size_t k = 0; k < nb; k++)
            for (int part = 0; part < 4; part++) {
                int g2 = 0;
                for (int i = 0; i < 8; i++) g2 += xq[k * 32 + (size_t)part * 8 + (size_t)i];
                xsum[k * 4 + (size_t)part] = g2;
            }
        #pragma omp parallel for schedule(static)
        for (long r = 0; r < (long)rows; r++) {
            float lanes[32];
            for (int l = 0; l < 32; l++) lanes[l] = 0.0f;
            const uint8_t *row = W + (size_t)r * nb * 20;
            for (size_t k = 0; k < nb; k++) {
                const uint8_t *b = row + k * 20;
                const float base = orc_h2f(rd16(b)), scale = orc_h2f(rd16(b + 2));
                const int32_t *xv = xq + k * 32;
                const float sx = xs[k];
                for (int part = 0; part < 4; part++) {
                    const uint8_t *c = b + 4 + part * 4;
                    int gs = 0;
                    for (int i = 0; i < 4; i++) gs += (c[i] & 0x0F) * xv[part * 8 + 2 * i] + (c[i] >> 4) * xv[part * 8 + 2 * i + 1];
                    const int lane = (int)(k % 8) * 4 + part;
                    float t = scale * (float)gs;
                    t = t * sx;
                    lanes[lane] = lanes[lane] + t;
                    float u = base * (float)xsum[k * 4 + (size_t)part];
                    u = u * sx;
                    lanes[lane] = lanes[lane] + u;
                }
            }
            for (int mask = 16; mask > 0; mask >>= 1) {
                float nv[32];
                for (int l = 0; l < 32; l++) nv[l] = lanes[l] + lanes[l ^ mask];
                memcpy(lanes, nv, sizeof(nv));
            }
            y[r] = orc_f2h(lanes[0]);
        }
        free(xsum); free(xq); free(xs);
        return 0;
    }
    #pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)rows; r++) {
        float lanes[32];
        int32_t q[64];
        double acc64 = 0.0;
        for (int l = 0; l < 32; l++) lanes[l] = 0.0f;
        for (size_t k = 0; k < nb; k++) {
            float scale, base;
            decode_block(dtype_w, W + ((size_t)r * nb + k) * (size_t)bb, q, &scale, &base);
            for (int part = 0; part < 4; part++) {
                int lane = (int)(k % 8) * 4 + part;
                int e0 = part * per_lane;
                size_t xblk = cap == 32 ? k : (2 * k + (size_t)(part >= 2));
                const int32_t *xv = xq + k * (size_t)cap + (size_t)e0;
                int gs = 0, gs2 = 0;
                for (int i = 0; i < per_lane; i++) { gs += q[e0 + i] * xv[i]; gs2 += xv[i]; }
                float t = scale * (float)gs;
                t = t * xs[xblk];
                lanes[lane] = lanes[lane] + t;
                if (dtype_w != ORC_Q8_B32T2) {
                    float u = base * (float)gs2;
                    u = u * xs[xblk];
                    lanes[lane] = lanes[lane] + u;
                }
                acc64 += ((double)scale * gs + (double)base * gs2) * (double)xs[xblk];
            }
        }
        for (int mask = 16; mask > 0; mask >>= 1) {
            float nv[32];
            for (int l = 0; l < 32; l++) nv[l] = lanes[l] + lanes[l ^ mask];
            memcpy(lanes, nv, sizeof(nv));
        }
        y[r] = orc_f2h(lanes[0]);
        if (y_f64) y_f64[r] = acc64;
    }
    free(xq); free(xs);
    return 0;
}

/* fp16-activation path: src/kernels/gemv.h:632-1497 (quantised weights are
 * dequantised to half, products accumulated in fp32) and :469-555 (F16
 * weights).  The reference accumulates Q8_B32T2 / Q4_B32T1 / F16 weights in
 * half (appendix A4); this restatement accumulates all of them in fp32 in
 * element order and also returns an fp64 value; tests compare with a stated
 * tolerance instead of reproducing the half-precision drift.              */
int orc_gemv_f16x(int dtype_w, const uint8_t *W, size_t rows, size_t cols,
                  const orc_f16 *x, const orc_f16 *bias, orc_f16 *y, double *y_f64)
{
    int cap = orc_block_capacity(dtype_w);
    if (cap <= 0 || cols % (size_t)cap != 0) return -1;
    size_t rb = orc_row_bytes(dtype_w, cols);
    float *xf = (float *)malloc(sizeof(float) * cols);
    if (!xf) return -1;
    orc_h2f_n(x, xf, cols);
    int err = 0;
    #pragma omp parallel
    {
        orc_f16 *wrow_h = (orc_f16 *)malloc(sizeof(orc_f16) * cols);
        #pragma omp for schedule(static)
        for (long r = 0; r < (long)rows; r++) {
            const orc_f16 *wh;
            if (dtype_w == ORC_F16) {
                wh = (const orc_f16 *)(W + (size_t)r * rb);
            } else if (dtype_w == ORC_F32) {
                orc_f2h_n((const float *)(W + (size_t)r * rb), wrow_h, cols);
                wh = wrow_h;
            } else {
                if (orc_dequantize_rows(dtype_w, W + (size_t)r * rb, 1, cols, wrow_h) != 0) { err = 1; continue; }
                wh = wrow_h;
            }
            float acc = 0.0f; double acc64 = 0.0;
            for (size_t c = 0; c < cols; c++) {
                float p = orc_h2f(wh[c]) * xf[c];
                acc = acc + p;
                acc64 += (double)p;
            }
            /* bias is a half add on the rounded result (gemv.h:524-526, TensorOpr::Add) */
            orc_f16 yh = orc_f2h(acc);
            if (bias) { yh = orc_f2h(orc_h2f(yh) + orc_h2f(bias[r])); acc64 += (double)orc_h2f(bias[r]); }
            y[r] = yh;
            if (y_f64) y_f64[r] = acc64;
        }
        free(wrow_h);
    }
    free(xf);
    return err ? -1 : 0;
}

/* The same product for T activation rows: Y[t] = orc_gemv_f16x(W, X[t]) for every t, bit for bit -- each weight row is
 * dequantised ONCE (the software F16 conversions of a row cost far more than the dot itself) and then multiplied with every
 * activation row by the loop of orc_gemv_f16x (same products, same serial fp32 order).  Test infrastructure only: what it buys
 * is the run time of the T > 1 steps of the whole-model oracle at 34B-40B widths. */
int orc_gemm_f16x(int dtype_w, const uint8_t *W, size_t rows, size_t cols, const orc_f16 *X, size_t T,
                  const orc_f16 *bias, orc_f16 *Y)
{
    int cap = orc_block_capacity(dtype_w);
    if (cap <= 0 || cols % (size_t)cap != 0 || T == 0) return -1;
    size_t rb = orc_row_bytes(dtype_w, cols);
    float *xf = (float *)malloc(sizeof(float) * cols * T);
    if (!xf) return -1;
    orc_h2f_n(X, xf, cols * T);
    int err = 0;
    #pragma omp parallel
    {
        orc_f16 *wrow_h = (orc_f16 *)malloc(sizeof(orc_f16) * cols);
        float *wf = (float *)malloc(sizeof(float) * cols);
        #pragma omp for schedule(static)
        for (long r = 0; r < (long)rows; r++) {
            const orc_f16 *wh;
            if (dtype_w == ORC_F16) {
                wh = (const orc_f16 *)(W + (size_t)r * rb);
            } else if (dtype_w == ORC_F32) {
                orc_f2h_n((const float *)(W + (size_t)r * rb), wrow_h, cols);
                wh = wrow_h;
            } else {
                if (orc_dequantize_rows(dtype_w, W + (size_t)r * rb, 1, cols, wrow_h) != 0) { err = 1; continue; }
                wh = wrow_h;
            }
            for (size_t c = 0; c < cols; c++) wf[c] = orc_h2f(wh[c]);
            for (size_t t = 0; t < T; t++) {
                const float *xt = xf + t * cols;
                float acc = 0.0f;
                for (size_t c = 0; c < cols; c++) {
                    float p = wf[c] * xt[c];
                    acc = acc + p;
                }
                orc_f16 yh = orc_f2h(acc);
                if (bias) yh = orc_f2h(orc_h2f(yh) + orc_h2f(bias[r]));
                Y[t * rows + (size_t)r] = yh;
            }
        }
        free(wrow_h); free(wf);
    }
    free(xf);
    return err ? -1 : 0;
}

/* ----------------------------------------------------------- normalisation */
/* Tensor_RmsNorm_Kernel, src/kernels/unary_tensor_opr.h:216-289; launcher
 * block (128,1), eps 1e-5 (src/tensor/tensor_opr.cu:568-577).             */
void orc_rmsnorm(const orc_f16 *x, size_t rows, size_t cols, const orc_f16 *w,
                 const orc_f16 *b, float multi_base, float eps, int nthreads_x, orc_f16 *y)
{
    int bx = nthreads_x > 0 ? nthreads_x : 128;
    int x_len = (int)((cols + (size_t)bx - 1) / (size_t)bx);
    for (size_t r = 0; r < rows; r++) {
        const orc_f16 *src = x + r * cols;
        float total = 0.0f;
        for (int tid = 0; tid < bx; tid++) {
            size_t xs0 = (size_t)tid * (size_t)x_len;
            size_t xe = (size_t)(tid + 1) * (size_t)x_len;
            if (xe > cols) xe = cols;
            float sum = 0.0f;
            for (size_t xi = xs0; xi < xe; xi++) {
                double v = (double)orc_h2f(src[xi]);
                sum = (float)((double)sum + v * v);
            }
            total = total + sum;
        }
        float mean = total / (float)cols;
        float scale = 1.0f / sqrtf(mean + eps);
        for (size_t xi = 0; xi < cols; xi++) {
            float v = orc_h2f(src[xi]) * scale;
            if (w && b) {
                float m = multi_base + orc_h2f(w[xi]);
                v = v * m;
                v = v + orc_h2f(b[xi]);
            } else if (w) {
                float m = multi_base + orc_h2f(w[xi]);
                v = v * m;
            }
            y[r * cols + xi] = orc_f2h(v);
        }
    }
}

/* Tensor_StdNorm_Kernel, src/kernels/unary_tensor_opr.h:68-149 */
void orc_stdnorm(const orc_f16 *x, size_t rows, size_t cols, const orc_f16 *w,
                 const orc_f16 *b, float eps, int nthreads_x, orc_f16 *y)
{
    int bx = nthreads_x > 0 ? nthreads_x : 128;
    for (size_t r = 0; r < rows; r++) {
        const orc_f16 *src = x + r * cols;
        float tsum = 0.0f, tsum2 = 0.0f;
        for (int tid = 0; tid < bx; tid++) {
            float sum = 0.0f, sum2 = 0.0f;
            for (size_t xi = (size_t)tid; xi < cols; xi += (size_t)bx) {
                double v = (double)orc_h2f(src[xi]);
                sum = (float)((double)sum + v);
                sum2 = (float)((double)sum2 + v * v);
            }
            tsum = tsum + sum; tsum2 = tsum2 + sum2;
        }
        float mean = tsum / (float)cols;
        float var = tsum2 / (float)cols - mean * mean;
        float scale = 1.0f / sqrtf(var + eps);
        for (size_t xi = 0; xi < cols; xi++) {
            float v = (orc_h2f(src[xi]) - mean) * scale;
            if (w && b) { v = v * orc_h2f(w[xi]); v = v + orc_h2f(b[xi]); }
            else if (w) { v = v * orc_h2f(w[xi]); }
            y[r * cols + xi] = orc_f2h(v);
        }
    }
}

/* ------------------------------------------------------ position embedding */
/* PosEmbedding_Rope_Std_Kernel :661-697, PosEmbedding_Rope_Order2_Kernel
 * :699-740 (src/kernels/unary_tensor_opr.h); dispatch and rope_dims/rope_cols
 * src/tensor/tensor_opr.cu:693-740 (F16 path, appendix A12).  x is
 * [tokens][heads][head_dim]; token t sits at position pos0 + t.           */
void orc_rope(orc_f16 *x, int head_dim, int heads, int tokens, int pos0,
              float theta, int order, int rope_dims, int rope_cols)
{
    if (order == 2) {
        const float theta_scale = powf(theta, -2.0f / (float)rope_dims);
        for (int t = 0; t < tokens; t++)
            for (int h = 0; h < heads; h++) {
                orc_f16 *row = x + ((size_t)t * (size_t)heads + (size_t)h) * (size_t)head_dim;
                for (int col = 0; 2 * col < head_dim; col++) {
                    float ang = (float)(pos0 + t);
                    if (col > 0) ang *= powf(theta_scale, (float)col);
                    float c = cosf(ang), s = sinf(ang);
                    if (2 * col < rope_cols) {
                        float x0 = orc_h2f(row[col]), x1 = orc_h2f(row[col + rope_cols / 2]);
                        float a = x0 * c, bq = x1 * s;
                        float d = x0 * s, e = x1 * c;
                        row[col] = orc_f2h(a - bq);
                        row[col + rope_cols / 2] = orc_f2h(d + e);
                    }
                }
            }
    } else {
        const float theta_scale = powf(theta, -2.0f / (float)rope_dims);
        for (int t = 0; t < tokens; t++)
            for (int h = 0; h < heads; h++) {
                orc_f16 *row = x + ((size_t)t * (size_t)heads + (size_t)h) * (size_t)head_dim;
                for (int col = 0; col + 1 < head_dim; col += 2) {
                    float ang = (float)(pos0 + t);
                    if (col > 0) ang *= powf(theta_scale, (float)(col / 2));
                    float c = cosf(ang), s = sinf(ang);
                    float x0 = orc_h2f(row[col]), x1 = orc_h2f(row[col + 1]);
                    float a = x0 * c, bq = x1 * s;
                    float d = x0 * s, e = x1 * c;
                    row[col] = orc_f2h(a - bq);
                    row[col + 1] = orc_f2h(d + e);
                }
            }
    }
}

/* PosEmbedding_Alibi_Std_Kernel :742-762; scores [heads][q_tokens][ctx].
 * base_head: head offset of this shard (appendix A13: the reference passes a
 * wrong offset under TP; the correct one is used here).                   */
void orc_alibi(orc_f16 *scores, int ctx, int q_tokens, int heads, int base_head, int total_heads)
{
    const int hl2 = 1 << (int)floor(log2((double)(float)total_heads));
    const float m0 = powf(2.0f, -8.0f / (float)hl2);
    const float m1 = powf(2.0f, -4.0f / (float)hl2);
    for (int h = 0; h < heads; h++) {
        int idx2 = h + base_head;
        float mk = idx2 < hl2 ? powf(m0, (float)(idx2 + 1)) : powf(m1, (float)(2 * (idx2 - hl2) + 1));
        for (int t = 0; t < q_tokens; t++) {
            orc_f16 *row = scores + ((size_t)h * (size_t)q_tokens + (size_t)t) * (size_t)ctx;
            for (int col = 0; col < ctx; col++) {
                float a = (float)col * mk;
                row[col] = orc_f2h(a + orc_h2f(row[col]));
            }
        }
    }
}

/* ----------------------------------------------------------------- softmax */
/* Tensor_SoftMax_Alg2_Kernel :480-535; block 32, grid (1,cy,cz)
 * (src/tensor/tensor_opr.cu:1189-1221).  s is [cz][cy][cx], in place.     */
void orc_softmax(orc_f16 *s, int cx, int cy, int cz, int prefix_len, float scale)
{
    for (int z = 0; z < cz; z++)
        for (int r = 0; r < cy; r++) {
            orc_f16 *row = s + ((size_t)z * (size_t)cy + (size_t)r) * (size_t)cx;
            float mx = -INFINITY;
            for (int xi = 0; xi < cx; xi++) {
                float v = scale * orc_h2f(row[xi]);
                if (prefix_len >= 0 && xi > prefix_len + r) v = -INFINITY;
                mx = mx > v ? mx : v;
            }
            float lanes[32];
            for (int l = 0; l < 32; l++) lanes[l] = 0.0f;
            for (int xi = 0; xi < cx; xi++) {
                float v = scale * orc_h2f(row[xi]);
                if (prefix_len >= 0 && xi > prefix_len + r) v = -INFINITY;
                float e = expf(v - mx);
                lanes[xi % 32] = lanes[xi % 32] + e;
                row[xi] = orc_f2h(e);
            }
            for (int mask = 16; mask > 0; mask >>= 1) {
                float nv[32];
                for (int l = 0; l < 32; l++) nv[l] = lanes[l] + lanes[l ^ mask];
                memcpy(lanes, nv, sizeof(nv));
            }
            float inv = 1.0f / lanes[0];
            for (int xi = 0; xi < cx; xi++) row[xi] = orc_f2h(orc_h2f(row[xi]) * inv);
        }
}

/* ------------------------------------------------------------- activations */
/* SiluActivation_Kernel :552-576, GeluActivation_Kernel :578-594,
 * ReluActivation_Kernel :537-550.  kind: 0 silu, 1 gelu, 2 relu.
 * With is_glu the input row is [2*n_cols] and out = act(x[:n])*x[n:].     */
void orc_act(const orc_f16 *x, size_t n_rows, size_t n_cols, int kind, int is_glu, orc_f16 *y)
{
    static const float GELU_COEF_A = 0.044715f;
    static const float SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    for (size_t r = 0; r < n_rows; r++)
        for (size_t c = 0; c < n_cols; c++) {
            size_t in_off = is_glu ? (r * 2 * n_cols + c) : (r * n_cols + c);
            float v = orc_h2f(x[in_off]);
            float fx;
            if (kind == 0) {
                fx = v / (1.0f + expf(-v));
                if (is_glu) fx = fx * orc_h2f(x[in_off + n_cols]);
            } else if (kind == 1) {
                float v2 = v * v;
                float inner = 1.0f + GELU_COEF_A * v2;
                float a = SQRT_2_OVER_PI * v;
                a = a * inner;
                float t = 1.0f + tanhf(a);
                fx = 0.5f * v;
                fx = fx * t;
            } else {
                fx = v > 0 ? v : 0;
            }
            y[r * n_cols + c] = orc_f2h(fx);
        }
}

/* ElementwiseAdd_Alg3_Half_Kernel, src/kernels/binary_tensor_opr.h:40-78:
 * half add = one rounding of the exact sum.  b repeats with period b_period. */
void orc_add(const orc_f16 *a, const orc_f16 *b, size_t n, size_t b_period, orc_f16 *c)
{
    for (size_t i = 0; i < n; i++) {
        size_t j = b_period ? (i % b_period) : i;
        c[i] = orc_f2h(orc_h2f(a[i]) + orc_h2f(b[j]));
    }
}
/* ElementwiseMul_Alg1_Kernel :127-140 */
void orc_mul(const orc_f16 *a, const orc_f16 *b, size_t n, orc_f16 *c)
{
    for (size_t i = 0; i < n; i++) c[i] = orc_f2h(orc_h2f(a[i]) * orc_h2f(b[i]));
}
/* Tensor_Scale_Kernel :142-153 */
void orc_scale(const orc_f16 *a, float s, size_t n, orc_f16 *c)
{
    for (size_t i = 0; i < n; i++) c[i] = orc_f2h(orc_h2f(a[i]) * s);
}

/* --------------------------------------------------------------- attention */
/* Per query: S = half(alpha * q.K) with alpha = 1/sqrt(hd)/kq_scale
 * (Gemm_Alg2_Kernel, src/kernels/gemm.h:83-178: fp32 accumulate in k order,
 * inference_worker.cc:1639-1724), optional ALiBi (:1183-1191), softmax with
 * causal mask and scale kq_scale (:1213), O = half(P.V) (:1257-1312).
 * Caches hold rows [n][kv_heads*head_dim] in F16 or Q8_B32T2 (kv_cache.cc:
 * 104-249; Q8 rows are dequantised exactly as q*scale -> half).
 * q: [q_tokens][heads][head_dim]; out: [q_tokens][heads*head_dim].        */
void orc_attention(const orc_f16 *q, const void *kcache, const void *vcache, int kv_dtype,
                   int n_ctx, int q_tokens, int prefix_len, int heads, int kv_heads,
                   int head_dim, float kq_scale, int use_alibi, int alibi_base_head,
                   int alibi_total_heads, orc_f16 *out)
{
    const int kv_dim = kv_heads * head_dim;
    const int group = heads / kv_heads;
    orc_f16 *K = NULL, *V = NULL;
    const orc_f16 *Kp, *Vp;
    if (kv_dtype == ORC_Q8_B32T2) {
        K = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)n_ctx * (size_t)kv_dim);
        V = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)n_ctx * (size_t)kv_dim);
        orc_dequantize_rows(ORC_Q8_B32T2, (const uint8_t *)kcache, (size_t)n_ctx, (size_t)kv_dim, K);
        orc_dequantize_rows(ORC_Q8_B32T2, (const uint8_t *)vcache, (size_t)n_ctx, (size_t)kv_dim, V);
        Kp = K; Vp = V;
    } else {
        Kp = (const orc_f16 *)kcache; Vp = (const orc_f16 *)vcache;
    }
    const float alpha = 1.0f / sqrtf((float)head_dim) / kq_scale;
    orc_f16 *S = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)heads * (size_t)q_tokens * (size_t)n_ctx);
    for (int h = 0; h < heads; h++) {
        int kvh = h / group;
        for (int t = 0; t < q_tokens; t++) {
            const orc_f16 *qv = q + ((size_t)t * (size_t)heads + (size_t)h) * (size_t)head_dim;
            orc_f16 *srow = S + ((size_t)h * (size_t)q_tokens + (size_t)t) * (size_t)n_ctx;
            for (int j = 0; j < n_ctx; j++) {
                const orc_f16 *kv = Kp + (size_t)j * (size_t)kv_dim + (size_t)kvh * (size_t)head_dim;
                float c = 0.0f;
                for (int d = 0; d < head_dim; d++) { float p = orc_h2f(qv[d]) * orc_h2f(kv[d]); c = c + p; }
                srow[j] = orc_f2h(alpha * c);
            }
        }
    }
    if (use_alibi) orc_alibi(S, n_ctx, q_tokens, heads, alibi_base_head, alibi_total_heads);
    orc_softmax(S, n_ctx, q_tokens, heads, prefix_len, kq_scale);
    for (int h = 0; h < heads; h++) {
        int kvh = h / group;
        for (int t = 0; t < q_tokens; t++) {
            const orc_f16 *prow = S + ((size_t)h * (size_t)q_tokens + (size_t)t) * (size_t)n_ctx;
            for (int d = 0; d < head_dim; d++) {
                float c = 0.0f;
                for (int j = 0; j < n_ctx; j++) {
                    float p = orc_h2f(prow[j]) * orc_h2f(Vp[(size_t)j * (size_t)kv_dim + (size_t)kvh * (size_t)head_dim + (size_t)d]);
                    c = c + p;
                }
                out[(size_t)t * (size_t)heads * (size_t)head_dim + (size_t)h * (size_t)head_dim + (size_t)d] = orc_f2h(1.0f * c);
            }
        }
    }
    free(S); free(K); free(V);
}

/* ------------------------------------------------------------- MoE routing */
/* HostTensorOpr::BuildRowsForMoE, src/tensor/host_tensor_opr.cc:190-244:
 * top-k of the router softmax row; experts with score < 1e-5 are skipped;
 * weights renormalised over the kept ones when norm_topk.  Returns count. */
int orc_moe_topk(const float *probs, int experts, int top_k, int norm_topk, int *idx, float *w)
{
    int n = 0;
    unsigned char *used = (unsigned char *)calloc((size_t)experts, 1);
    for (int k = 0; k < top_k && k < experts; k++) {
        int best = -1;
        for (int e = 0; e < experts; e++) {
            if (used[e]) continue;
            if (best < 0 || probs[e] > probs[best]) best = e;
        }
        if (best < 0) break;
        used[best] = 1;
        if (probs[best] < 0.00001f) continue;
        idx[n] = best; w[n] = probs[best]; n++;
    }
    if (norm_topk && n > 0) {
        float sum = 0.0f;
        for (int i = 0; i < n; i++) sum = sum + w[i];
        for (int i = 0; i < n; i++) w[i] = w[i] / sum;
    }
    free(used);
    return n;
}
