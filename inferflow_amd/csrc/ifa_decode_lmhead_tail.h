// ifa_decode_lmhead_tail.h -- round 4: the END of a decode step as ONE launch: final norm + F16 lm_head GEMV (k_dec_lmhead_f16's
// body) + greedy argmax + advance of the device state (k_dec_argmax_advance) + the NEXT step's embedding gather and RoPE table
// (k_dec_gather).  Two launches less per token: both were pure latency chains (8.5 + 5.0 us for 64 KB of logits and an 8 KB row).
//
// Every wave already holds the logits of its rows (lane 0 after the wave sum): it tracks its best (value, id) by the rules of
// argmax_scan (largest value, lowest id among equals, the excluded ids never offered: SamplingStrategy::GetSortedTopK,
// sampling_strategy.cc:281-297), the workgroup reduces its 8 waves, stores ONE 64-bit key -- (ordered value bits << 32) |
// (0xFFFFFFFF - id): an unsigned maximum is the argmax -- past the caches, drains the store and bumps a counter; the workgroup
// whose bump is the launch's last (old == grid - 1; it puts the counter back to zero for the next launch) reads all keys,
// writes the state words exactly as k_dec_argmax_advance does, copies the new token's embedding row and fills the RoPE table of
// the new position.  Same logits buffer, same ids, same state: tokens bit-identical to the three-launch tail (tests).
#pragma once
#include "ifa_decode_kernels.h"

namespace ifa {

struct DecStepTail {
    int *state; int ring;
    unsigned long long *keys;       // [grid] per-workgroup best
    unsigned *counter;              // arrivals of the running launch (zero between launches)
    const half_t *embd; int vocab; half_t *x_out;
    float *rope_tab; int head_dim; float theta; int rope_dims; float embd_scale;
};

__device__ __forceinline__ uint32_t tail_ordered(float f)
{
    const uint32_t b = f == 0.0f ? 0u : __builtin_bit_cast(uint32_t, f);      // -0 == +0 in the scan's comparison: one key for both
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

template <int NJ, int R, int NORM>
__global__ void __launch_bounds__(DEC_THREADS) k_dec_lmhead_tail(const DecLmHeadParams P, const DecStepTail Z)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *xn = reinterpret_cast<half_t *>(smem);                         // [cols]
    float *part = reinterpret_cast<float *>(smem + (((size_t)P.cols * 2 + 15) & ~(size_t)15));      // [132]; [64..] reused by the tail
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = blockIdx.x * DEC_WAVES + wave;
    const int W = gridDim.x * DEC_WAVES;
    const int nbatch = (P.rows + R - 1) / R;
    const int chunks = P.cols >> 3;
    const __attribute__((address_space(4))) int *cs = (const __attribute__((address_space(4))) int *)Z.state;
    const int st_pos = cs[1], st_step = cs[2];
    const int ne = min(max(cs[3], 0), 3);
    const int x4 = cs[4], x5 = cs[5], x6 = cs[6];
    const int e0 = ne > 0 ? x4 : -1, e1 = ne > 1 ? x5 : -1, e2 = ne > 2 ? x6 : -1;

    auto load_batch = [&](u32x4 (&dst)[R][NJ], int b) {
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const int row = min(b * R + rr, P.rows - 1);
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int c = min(lane + 64 * j, chunks - 1);
                dst[rr][j] = nt_load<u32x4>(reinterpret_cast<const u32x4 *>(P.W + (size_t)row * P.cols) + c);
            }
        }
    };
    u32x4 cur[R][NJ];
    load_batch(cur, min(gw, nbatch - 1));

    for (int c = tid; c < chunks; c += DEC_THREADS)
        *reinterpret_cast<half8_t *>(xn + (size_t)c * 8) = *reinterpret_cast<const half8_t *>(P.x + (size_t)c * 8);
    if constexpr (NORM == 1) {
        const int ngroups = (chunks + 63) >> 6;
        __syncthreads();
        for (int g = tid >> 6; g < ngroups; g += DEC_WAVES) {
            const int c = 64 * g + lane;
            half8_t v8;
#pragma unroll
            for (int i = 0; i < 8; i++) v8[i] = (half_t)0;
            if (c < chunks) v8 = *reinterpret_cast<const half8_t *>(xn + (size_t)c * 8);
            const float pg = wave_sum(rms_chunk_sq(v8));
            if (lane == 0) part[g] = pg;
        }
        __syncthreads();
        const float scale = rms_scale_of(rms_total(part, ngroups), P.cols, P.eps);
        for (int c = tid; c < chunks; c += DEC_THREADS) {
            half8_t xv = *reinterpret_cast<const half8_t *>(xn + (size_t)c * 8);
            half8_t wv, bv;
            if (P.norm_w) wv = *reinterpret_cast<const half8_t *>(P.norm_w + (size_t)c * 8);
            if (P.norm_b) bv = *reinterpret_cast<const half8_t *>(P.norm_b + (size_t)c * 8);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float t = (float)xv[i] * scale;
                if (P.norm_w) {
                    float m = P.multi_base + (float)wv[i];
                    t = t * m;
                    if (P.norm_b) t = t + (float)bv[i];
                }
                xv[i] = f2h(t);
            }
            *reinterpret_cast<half8_t *>(xn + (size_t)c * 8) = xv;
            if (P.xn_out && blockIdx.x == 0) *reinterpret_cast<half8_t *>(P.xn_out + (size_t)c * 8) = xv;
        }
    } else {
        if (P.xn_out && blockIdx.x == 0)
            for (int c = tid; c < chunks; c += DEC_THREADS)
                *reinterpret_cast<half8_t *>(P.xn_out + (size_t)c * 8) = *reinterpret_cast<const half8_t *>(xn + (size_t)c * 8);
    }
    __syncthreads();
    u32x4 xr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int c = lane + 64 * j;
        xr[j] = u32x4{0, 0, 0, 0};
        if (c < chunks) xr[j] = *reinterpret_cast<const u32x4 *>(xn + (size_t)c * 8);
    }
    float best = -INFINITY; int besti = 0x7FFFFFFF;       // lane 0 of the wave: its rows' best, argmax_scan's rule
    for (int b = gw; b < nbatch; b += W) {
        u32x4 nxt[R][NJ];
        load_batch(nxt, b + W);
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; j++) acc = dot8_f16(cur[rr][j], xr[j], acc);
            acc = wave_sum(acc);
            const int row = b * R + rr;
            if (lane == 0 && row < P.rows) {
                const half_t y = f2h(acc);
                P.logits[row] = y;
                const float f = (float)y;
                if (row != e0 && row != e1 && row != e2 && (f > best || (f == best && row < besti))) { best = f; besti = row; }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; rr++)
#pragma unroll
            for (int j = 0; j < NJ; j++) cur[rr][j] = nxt[rr][j];
    }
    // ---- the workgroup's best: 8 (value, id) pairs through LDS (the prologue's scratch is free)
    float *bv = part + 64; int *bi = reinterpret_cast<int *>(part + 80);
    unsigned *flag = reinterpret_cast<unsigned *>(part + 96);
    if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < DEC_WAVES; w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        const unsigned long long key = besti == 0x7FFFFFFF ? 0ull : (((unsigned long long)tail_ordered(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)besti));
        __hip_atomic_store(Z.keys + blockIdx.x, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the key is in memory before the arrival is counted
        const unsigned old = __hip_atomic_fetch_add(Z.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = (old == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (flag[0] == 0u) return;
    // ---- the launch's last workgroup: the argmax over the workgroups' keys, then k_dec_argmax_advance's state update and
    // k_dec_gather's row copy + RoPE table for the NEW token at the NEW position
    if (tid == 0) __hip_atomic_store(Z.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every arrival of this launch is in: the next launch counts from zero
    unsigned long long mk = 0ull;
    for (int i = tid; i < (int)gridDim.x; i += DEC_THREADS) {
        const unsigned long long k = __hip_atomic_load(Z.keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mk = k > mk ? k : mk;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)mk, m), hi = __shfl_xor((uint32_t)(mk >> 32), m);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        mk = o > mk ? o : mk;
    }
    unsigned long long *wk = reinterpret_cast<unsigned long long *>(part + 100);       // [8], 8-byte aligned: part is 16-byte aligned
    if (lane == 0) wk[wave] = mk;
    __syncthreads();
    int *tokp = reinterpret_cast<int *>(part + 120);
    if (tid == 0) {
        for (int w = 1; w < DEC_WAVES; w++) mk = wk[w] > mk ? wk[w] : mk;
        const int tok = mk == 0ull ? 0 : (int)(0xFFFFFFFFu - (uint32_t)mk);
        Z.state[8 + (st_step % Z.ring)] = tok;
        Z.state[0] = tok;
        Z.state[1] = st_pos + 1;
        Z.state[2] = st_step + 1;
        tokp[0] = tok;
    }
    __syncthreads();
    if (Z.embd) {
        const int tok = min(max(tokp[0], 0), Z.vocab - 1);
        for (int c = tid; c < chunks; c += DEC_THREADS)
            reinterpret_cast<u32x4 *>(Z.x_out)[c] = embd_row_scale(reinterpret_cast<const u32x4 *>(Z.embd + (size_t)tok * P.cols)[c], Z.embd_scale);
    }
    if (Z.rope_tab) {
        const int pos = st_pos + 1;
        for (int c = tid; c < Z.head_dim / 2; c += DEC_THREADS) {
            float cs2, sn;
            rope_angle(c, pos, Z.theta, Z.rope_dims, cs2, sn);
            Z.rope_tab[2 * c] = cs2; Z.rope_tab[2 * c + 1] = sn;
        }
    }
}

} // namespace ifa
