// ifa_gemm.hip -- prefill / batched linear layer:  Y[T][N] = X[T][K] . W[N][K]^T (+bias)
//
// Reference (MatrixMultiplication, src/transformer/inference_worker.cc:2374-2415):
// dequantise the WHOLE weight tensor to F16 scratch (TensorOpr::Dequantize), cublasGemmEx
// F16 x F16 -> F16 with fp32 accumulation (src/tensor/cublas_engine.cu:420-436), transpose.
// Here the dequantisation is fused into the GEMM: every lane decodes the quant block it needs
// straight into its MFMA B-operand registers (values rounded to half exactly like the
// reference's dequant tensor), the activation tile is staged through LDS, products go to
// v_mfma_f32_32x32x16_f16 (fp32 accumulate), one F16 rounding at the end, bias as a half add.
// No full-tensor F16 copy is written or re-read, and no transpose.
//
// MFMA 32x32x16 operand mapping (cdna_hip_programming.md §3): lane (i = lane&31, g = lane>>5)
// holds A[i][8g..8g+7] and B[8g..8g+7][i]; D: col = lane&31, row = (r&3) + 8*(r>>2) + 4*g.
// Tokens are the A rows, weight rows the B columns.  k is consumed in a permuted order
// (lane half g works through quant block 2*step+g): the sum over k does not care.
#include <algorithm>
#include <type_traits>
#include "ifa_host.h"
#include "ifa_codec.h"
#include "ifa_moe.h"

namespace ifa {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_THREADS = 256;     // 4 waves, 32 weight rows each
constexpr int GEMM_ROWS = 128;

// weights of quant block `b` of row `n` as CAP halfs (reference rounding: half(q*scale+base))
template <int DT, int CAP>
__device__ __forceinline__ void load_block_f16(const uint8_t *__restrict__ W, size_t row, int nblk, int b, bool ok,
                                               half_t (&v)[CAP])
{
    if constexpr (DT == F16) {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(W + (row * (size_t)nblk + (size_t)b) * (CAP * 2));
#pragma unroll
        for (int i = 0; i < CAP / 8; i++) {
            u32x4 t = ok ? p[i] : u32x4{0, 0, 0, 0};
            const half8_t h = __builtin_bit_cast(half8_t, t);
#pragma unroll
            for (int e = 0; e < 8; e++) v[8 * i + e] = h[e];
        }
    } else if constexpr (DT == Q4_B32T1A || DT == Q4_B32T1B) {
        // 20-byte blocks are 4-byte aligned: five dword loads, q*scale+base as one fp32 fma
        // (q*scale is exact in fp32, so the fma rounds exactly like the reference's mul + add)
        const uint32_t *p = reinterpret_cast<const uint32_t *>(W + (row * (size_t)nblk + (size_t)b) * 20);
        const uint32_t sb = p[0];
        const float base = hbits2f((uint16_t)(sb & 0xFFFFu)), scale = hbits2f((uint16_t)(sb >> 16));
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t c = p[1 + w];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float q = (float)((c >> (4 * e)) & 0xFu);
                v[8 * w + e] = ok ? f2h(__builtin_fmaf(q, scale, base)) : (half_t)0;
            }
        }
    } else {
        constexpr int BB = block_bytes(DT);
        RawBlock<BB> blk;
        blk.load(W + (row * (size_t)nblk + (size_t)b) * BB);
        int q[CAP]; float scale, base;
        decode_block<DT>(blk, q, scale, base);
#pragma unroll
        for (int i = 0; i < CAP; i++) v[i] = ok ? f2h(block_value<DT>(q[i], scale, base)) : (half_t)0;
    }
}

// the same block in two steps -- raw bytes now, halfs later -- so that the next K step's loads are in flight
// while the current one is multiplied
template <int DT, int CAP>
struct WRaw {
    static constexpr int BB = (DT == F16) ? CAP * 2 : block_bytes(DT);
    static constexpr bool Q4FAST = (DT == Q4_B32T1A || DT == Q4_B32T1B);
    u32x4 f16v[(DT == F16) ? CAP / 8 : 1];
    uint32_t q4[Q4FAST ? 5 : 1];
    RawBlock<(DT == F16 || Q4FAST) ? 2 : BB> blk;
    __device__ __forceinline__ void load(const uint8_t *__restrict__ W, size_t row, int nblk, int b)
    {
        const uint8_t *p = W + (row * (size_t)nblk + (size_t)b) * BB;
        if constexpr (DT == F16) {
#pragma unroll
            for (int i = 0; i < CAP / 8; i++) f16v[i] = reinterpret_cast<const u32x4 *>(p)[i];
        } else if constexpr (Q4FAST) {
#pragma unroll
            for (int i = 0; i < 5; i++) q4[i] = reinterpret_cast<const uint32_t *>(p)[i];
        } else {
            blk.load(p);
        }
    }
    __device__ __forceinline__ void decode(bool ok, half_t (&v)[CAP]) const
    {
        if constexpr (DT == F16) {
#pragma unroll
            for (int i = 0; i < CAP / 8; i++) {
                const half8_t h = __builtin_bit_cast(half8_t, ok ? f16v[i] : u32x4{0, 0, 0, 0});
#pragma unroll
                for (int e = 0; e < 8; e++) v[8 * i + e] = h[e];
            }
        } else if constexpr (Q4FAST) {
            // masked blocks: scale = base = 0 gives exact zeros (two selects instead of one per value); nibbles are spread
            // into bytes first so that each value is ONE v_cvt_f32_ubyteN instead of a bit-field extract + convert
            const float base = ok ? hbits2f((uint16_t)(q4[0] & 0xFFFFu)) : 0.0f, scale = ok ? hbits2f((uint16_t)(q4[0] >> 16)) : 0.0f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint32_t lo = q4[1 + w] & 0x0F0F0F0Fu, hi = (q4[1 + w] >> 4) & 0x0F0F0F0Fu;
                const float ql[4] = {ubyte_f32<0>(lo), ubyte_f32<1>(lo), ubyte_f32<2>(lo), ubyte_f32<3>(lo)};
                const float qh[4] = {ubyte_f32<0>(hi), ubyte_f32<1>(hi), ubyte_f32<2>(hi), ubyte_f32<3>(hi)};
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    v[8 * w + 2 * b] = f2h(__builtin_fmaf(ql[b], scale, base));        // q*scale exact: fma == mul + add
                    v[8 * w + 2 * b + 1] = f2h(__builtin_fmaf(qh[b], scale, base));
                }
            }
        } else {
            int q[CAP]; float scale, base;
            decode_block<DT>(blk, q, scale, base);
            if (!ok) { scale = 0.0f; base = 0.0f; }          // codes are finite integers: exact zeros, no per-value select
#pragma unroll
            for (int i = 0; i < CAP; i++) v[i] = f2h(block_value<DT>(q[i], scale, base));
        }
    }
};

// SPLITK = false: 4 waves x 32 rows per workgroup, all waves walk the whole K (large T).
// SPLITK = true : the 4 waves share ONE 32-row tile and take every 4th K step each (their own
//                 LDS slab, no workgroup barrier in the loop), partial tiles summed through LDS at
//                 the end: 4x more workgroups when T is small and the layer is weight-stream bound.
// grp.on: grouped launch (mixture of experts, ifa_moe.h): blockIdx.y is a tile of <= 32*MT rows of ONE expert -- its
// weights come from the pointer table, its activation / output rows start at the tile's entry offset
template <int DT, int MT, bool SPLITK, int NW = 4>
__global__ void __launch_bounds__(NW * 64) k_gemm_q(const uint8_t *__restrict__ W, int N, int nblk,
                                                         const half_t *__restrict__ X, int T, int K,
                                                         const half_t *__restrict__ bias, half_t *__restrict__ Y, const MoeGroup grp = MoeGroup())
{
    constexpr int CAP = (DT == F16) ? 32 : block_capacity(DT);
    constexpr int KSTEP = 2 * CAP;                 // one quant block per lane half and step
    constexpr int XROW = KSTEP * 2 + 16;           // bytes per staged activation row (+16: bank spread)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(SPLITK || NW == 4, "the non-split variant is 4 waves x 32 rows");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int n = SPLITK ? blockIdx.x * 32 + i : blockIdx.x * GEMM_ROWS + wave * 32 + i;
    const size_t nrow = (size_t)min(n, N - 1);
    char *slab = SPLITK ? smem + (size_t)wave * (32 * MT * XROW) : smem;
    int t0 = blockIdx.y * 32 * MT;
    if (grp.on) {
        if ((int)blockIdx.y >= grp.counts[1]) return;
        const MoeTile tl = grp.tiles[blockIdx.y];
        W = grp.wtab[tl.expert * 3 + grp.which];
        X += (size_t)tl.row0 * K; Y += (size_t)tl.row0 * N;
        T = tl.nrows; t0 = 0;
    }
    f32x16_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[mt][r] = 0.0f;

    const int nsteps = (nblk + 1) / 2;
    constexpr int CHUNKS_PER_ROW = KSTEP / 8;
    if constexpr (!SPLITK) {
        // ---- software-pipelined: the activation chunks and the weight block of step s+1 are requested before
        // step s is multiplied (one LDS buffer, registers carry the next step)
        constexpr int NCH = 32 * MT * CHUNKS_PER_ROW / GEMM_THREADS;      // 16-byte activation chunks per thread and step
        u32x4 xa[NCH];
        WRaw<DT, CAP> wr;
        auto fetch = [&](int step) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int idx = tid + c * GEMM_THREADS;
                const int r = idx / CHUNKS_PER_ROW, cc = idx % CHUNKS_PER_ROW;
                const int tok = min(t0 + r, T - 1), k = min(step * KSTEP + cc * 8, K - 8);
                xa[c] = *reinterpret_cast<const u32x4 *>(X + (size_t)tok * K + k);   // clamped, masked when stored
            }
            wr.load(W, nrow, nblk, min(2 * step + g, nblk - 1));
        };
        fetch(0);
        for (int step = 0; step < nsteps; step++) {
            __syncthreads();                         // the previous step's fragments have been read
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int idx = tid + c * GEMM_THREADS;
                const int r = idx / CHUNKS_PER_ROW, cc = idx % CHUNKS_PER_ROW;
                const bool ok = (t0 + r < T) && (step * KSTEP + cc * 8 < K);
                *reinterpret_cast<u32x4 *>(smem + (size_t)r * XROW + (size_t)cc * 16) = ok ? xa[c] : u32x4{0, 0, 0, 0};
            }
            half_t v[CAP];
            wr.decode(2 * step + g < nblk, v);
            if (step + 1 < nsteps) fetch(step + 1);
            __syncthreads();
#pragma unroll
            for (int m = 0; m < CAP / 8; m++) {
                half8_t bfrag;
#pragma unroll
                for (int e = 0; e < 8; e++) bfrag[e] = v[8 * m + e];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const half8_t afrag = *reinterpret_cast<const half8_t *>(smem + (size_t)(mt * 32 + i) * XROW
                                                                           + (size_t)(g * CAP + 8 * m) * 2);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag, bfrag, acc[mt], 0, 0, 0);
                }
            }
        }
    } else {
        // ---- split-K (small T): wave w owns K steps w, w+4, ...; NST register sets keep the loads of its next NST
        // steps in flight while one step is multiplied: the layer is a weight stream here, and what a CU can stream is
        // (bytes in flight) / latency -- two sets left 10 KB per CU in flight, 0.66 TB/s for the whole chip
        constexpr int NCHW = 32 * MT * CHUNKS_PER_ROW / 64;               // activation chunks per lane and step
        constexpr int NST = 4;
        struct Stage { u32x4 xa[NCHW]; WRaw<DT, CAP> wr; };
        Stage st[NST];
        auto fetch = [&](Stage &sg, int step) {
#pragma unroll
            for (int c = 0; c < NCHW; c++) {
                const int idx = lane + c * 64;
                const int r = idx / CHUNKS_PER_ROW, cc = idx % CHUNKS_PER_ROW;
                const int tok = min(t0 + r, T - 1), k = min(step * KSTEP + cc * 8, K - 8);
                sg.xa[c] = *reinterpret_cast<const u32x4 *>(X + (size_t)tok * K + k);
            }
            sg.wr.load(W, nrow, nblk, min(2 * step + g, nblk - 1));
        };
        // every load is unconditional (clamped step, masked when consumed): a load under a branch makes the compiler wait
        // vmcnt(0) at the join, which drains the whole ring at every step
        auto consume = [&](const Stage &sg, int step, bool valid) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < NCHW; c++) {
                const int idx = lane + c * 64;
                const int r = idx / CHUNKS_PER_ROW, cc = idx % CHUNKS_PER_ROW;
                const bool ok = valid && (t0 + r < T) && (step * KSTEP + cc * 8 < K);
                *reinterpret_cast<u32x4 *>(slab + (size_t)r * XROW + (size_t)cc * 16) = ok ? sg.xa[c] : u32x4{0, 0, 0, 0};
            }
            half_t v[CAP];
            sg.wr.decode(valid && 2 * step + g < nblk, v);
            __builtin_amdgcn_wave_barrier();     // one wave's LDS ops are ordered
#pragma unroll
            for (int m = 0; m < CAP / 8; m++) {
                half8_t bfrag;
#pragma unroll
                for (int e = 0; e < 8; e++) bfrag[e] = v[8 * m + e];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const half8_t afrag = *reinterpret_cast<const half8_t *>(slab + (size_t)(mt * 32 + i) * XROW
                                                                           + (size_t)(g * CAP + 8 * m) * 2);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag, bfrag, acc[mt], 0, 0, 0);
                }
            }
        };
        const int last = nsteps - 1;
        const int rounds = ((nsteps - wave + NW - 1) / NW + NST - 1) / NST;      // this wave's steps, in rounds of NST
#pragma unroll
        for (int u = 0; u < NST; u++) fetch(st[u], min(wave + NW * u, last));
        for (int r = 0; r < rounds; r++) {
#pragma unroll
            for (int u = 0; u < NST; u++) {
                const int su = wave + NW * (r * NST + u);
                consume(st[u], su, su < nsteps);
                fetch(st[u], min(su + NW * NST, last));
            }
        }
    }
    if constexpr (SPLITK) {     // sum the NW partial tiles: wave w parks its tile, wave 0 adds them in order 0,1,2,...
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem);      // [NW - 1][MT][16][64]
        if (wave > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) red[(((wave - 1) * MT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 0; w < NW - 1; w++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mt][r] = acc[mt][r] + red[((w * MT + mt) * 16 + r) * 64 + lane];
    }
    if (n >= N) return;
    const float bv = bias ? h2f(bias[n]) : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int tok = t0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (tok < T) {
                half_t y = f2h(acc[mt][r]);
                if (bias) y = f2h(h2f(y) + bv);
                Y[(size_t)tok * N + n] = y;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Large T (prefill, MFMA-bound): 128 tokens x BN (256 / 128) weight rows per workgroup, K walked in steps of 64.
//  * the weight tile is dequantised ONCE per workgroup and step into LDS (values rounded to half exactly like the
//    reference's Dequantize tensor) and shared by the MFMA waves -- the small-T kernel above decodes a block per lane
//    and MFMA tile, which bounds it at ~0.5 PFLOP/s (VALU per MFMA);
//  * wave specialisation: waves 0-3 only multiply (a 2 x 2 grid of 64-token x BN/2-row sub-tiles, fragments from LDS),
//    waves 4-7 only load and dequantise the NEXT step's tiles into the other LDS buffer.  The matrix pipe and the
//    VALU of a SIMD run side by side (MI355X_MICROARCH.md "Wave scheduling"), so a step costs max(MFMA, dequantisation)
//    instead of their sum; one barrier per step hands the buffers over;
//  * rows of both LDS tiles are padded to 144 bytes: the 16-byte fragment reads of 16 consecutive lanes hit 16 disjoint
//    groups of 4 banks (measured: SQ_LDS_BANK_CONFLICT = 0).
// Same arithmetic as k_gemm_q: fp32 accumulation of half products, one F16 rounding, bias as a half add.
constexpr int BIG_BM = 128, BIG_BK = 64, BIG_ROWB = BIG_BK * 2 + 16;     // LDS row stride in bytes

template <int DT, int BN>
__global__ void __launch_bounds__(512) k_gemm_big(const uint8_t *__restrict__ W, int N, int nblk,
                                                  const half_t *__restrict__ X, int T, int K,
                                                  const half_t *__restrict__ bias, half_t *__restrict__ Y, int dbg)
{
    constexpr int CAP = (DT == F16) ? 32 : block_capacity(DT);
    constexpr int BPS = BIG_BK / CAP;                   // quant blocks per row and step (2 or 1)
    constexpr int WB = (BN * BPS + 255) / 256;          // blocks a loader thread dequantises per step
    constexpr int NT = BN / 64;                         // 32-row accumulator tiles per MFMA wave along N (4 or 2)
    constexpr size_t TILE = (size_t)(BIG_BM + BN) * BIG_ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BN, t0 = blockIdx.y * BIG_BM;
    const int nsteps = (K + BIG_BK - 1) / BIG_BK;
    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int lt = tid - 256;
        u32x4 xa[4];
        WRaw<DT, CAP> wr[WB];
        auto fetch = [&](int step) {
#pragma unroll
            for (int c = 0; c < 4; c++) {               // 128 rows x 8 chunks of 8 halfs
                const int idx = lt + c * 256;
                const int r = idx >> 3, cc = idx & 7;
                const int tok = min(t0 + r, T - 1), k = min(step * BIG_BK + cc * 8, K - 8);
                xa[c] = *reinterpret_cast<const u32x4 *>(X + (size_t)tok * K + k);
            }
#pragma unroll
            for (int j = 0; j < WB; j++) {
                const int idx = min(lt + j * 256, BN * BPS - 1);
                const int nl = idx / BPS, b = idx % BPS;
                wr[j].load(W, (size_t)min(n0 + nl, N - 1), nblk, min(step * BPS + b, nblk - 1));
            }
        };
        auto stage = [&](int step) {                    // registers -> LDS buffer (step & 1)
            char *As = smem + (size_t)(step & 1) * TILE, *Bs = As + (size_t)BIG_BM * BIG_ROWB;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int idx = lt + c * 256;
                const int r = idx >> 3, cc = idx & 7;
                const bool ok = (t0 + r < T) && (step * BIG_BK + cc * 8 < K);
                *reinterpret_cast<u32x4 *>(As + (size_t)r * BIG_ROWB + (size_t)cc * 16) = ok ? xa[c] : u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int j = 0; j < WB; j++) {
                const int idx = lt + j * 256;
                if (idx >= BN * BPS) continue;
                const int nl = idx / BPS, b = idx % BPS;
                if (dbg & 1) continue;                  // (ablation: no dequantisation, no B stores)
                half_t v[CAP];
                wr[j].decode(step * BPS + b < nblk, v);
#pragma unroll
                for (int m = 0; m < CAP / 8; m++) {
                    half8_t h;
#pragma unroll
                    for (int e = 0; e < 8; e++) h[e] = v[8 * m + e];
                    *reinterpret_cast<half8_t *>(Bs + (size_t)nl * BIG_ROWB + (size_t)(b * CAP + 8 * m) * 2) = h;
                }
            }
        };
        fetch(0);
        stage(0);
        if (nsteps > 1) fetch(1);
        __syncthreads();                                // tile 0 is ready
        for (int step = 0; step < nsteps; step++) {
            // the MFMA waves multiply tile `step`; tile step + 1 goes into the other buffer (its previous content,
            // tile step - 1, was released by the barrier that ended the previous iteration)
            if (step + 1 < nsteps) {
                stage(step + 1);
                if (step + 2 < nsteps) fetch(step + 2);
            }
            __syncthreads();
        }
        return;
    }
    // ---------------------------------------------------------------------- MFMA waves
    const int i = lane & 31, g = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;            // this wave's 64-token x (BN/2)-row sub-tile
    f32x16_t acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;
    __syncthreads();                                    // tile 0 is ready
    for (int step = 0; step < nsteps; step++) {
        const char *As = smem + (size_t)(step & 1) * TILE, *Bs = As + (size_t)BIG_BM * BIG_ROWB;
#pragma unroll
        for (int ks = 0; ks < BIG_BK / 16; ks++) {
            if (dbg & 2) continue;                      // (ablation: no fragment reads, no MFMAs)
            half8_t af[2], bf[NT];
#pragma unroll
            for (int a = 0; a < 2; a++)
                af[a] = *reinterpret_cast<const half8_t *>(As + (size_t)(wm * 64 + a * 32 + i) * BIG_ROWB + (size_t)(ks * 16 + 8 * g) * 2);
#pragma unroll
            for (int b = 0; b < NT; b++)
                bf[b] = *reinterpret_cast<const half8_t *>(Bs + (size_t)(wn * (BN / 2) + b * 32 + i) * BIG_ROWB + (size_t)(ks * 16 + 8 * g) * 2);
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < NT; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();                                // this tile may be overwritten, the next one is complete
    }
#pragma unroll
    for (int b = 0; b < NT; b++) {
        const int n = n0 + wn * (BN / 2) + b * 32 + i;
        if (n >= N) continue;
        const float bv = bias ? h2f(bias[n]) : 0.0f;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int tok = t0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (tok < T) {
                    half_t y = f2h(acc[a][b][r]);
                    if (bias) y = f2h(h2f(y) + bv);
                    Y[(size_t)tok * N + n] = y;
                }
            }
    }
}

} // namespace ifa

using namespace ifa;

namespace ifa {
bool gemm_lt_wanted(size_t tokens);                  // ifa_gemm_lt.hip
int gemm_lt(int w_dtype, const void *W, size_t N, size_t K, const void *X, size_t T, const void *bias, void *Y, hipStream_t s);
}

static int gemm_num_cus()
{
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t prop; n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
    return n;
}
static int g_gemm_big = 0;       // ifa_gemm_big_tiles(1): opt-in large-tile kernel for T > 128 (round 2: 0.44-0.52 PFLOP/s, behind the other routes; its load skeleton alone costs 57 % of its time -- tools/probes/gemm_big_ablation.py)

template <int DT>
static int launch_gemm(const void *W, size_t N, size_t K, const void *X, size_t T, const void *bias, void *Y, hipStream_t s)
{
    constexpr int CAP = (DT == F16) ? 32 : block_capacity(DT);
    const int nblk = (int)(K / CAP);
    constexpr int MT = 2;
    const size_t slab = (size_t)32 * MT * (2 * CAP * 2 + 16);
    if (T > 128 && K % 8 == 0 && g_gemm_big) {
        // MFMA-bound: the weight tile dequantised once per workgroup into LDS by loader waves, 128 x 256 output tiles
        // (128 x 128 when the wider tile would leave compute units without a workgroup)
        const bool wide = (size_t)ifa_cdiv(N, 256) * ifa_cdiv(T, BIG_BM) >= (size_t)gemm_num_cus();
        if (wide) {
            dim3 grid(ifa_cdiv(N, 256), ifa_cdiv(T, BIG_BM));
            const size_t smem = 2 * (size_t)(BIG_BM + 256) * BIG_ROWB;
            static bool attr_set = false;
            if (!attr_set) { (void)hipFuncSetAttribute((const void *)k_gemm_big<DT, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_set = true; }
            k_gemm_big<DT, 256><<<grid, dim3(512), smem, s>>>((const uint8_t *)W, (int)N, nblk, (const half_t *)X, (int)T, (int)K,
                                                              (const half_t *)bias, (half_t *)Y, g_gemm_big >> 4);
        } else {
            dim3 grid(ifa_cdiv(N, 128), ifa_cdiv(T, BIG_BM));
            const size_t smem = 2 * (size_t)(BIG_BM + 128) * BIG_ROWB;
            static bool attr_set = false;
            if (!attr_set) { (void)hipFuncSetAttribute((const void *)k_gemm_big<DT, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_set = true; }
            k_gemm_big<DT, 128><<<grid, dim3(512), smem, s>>>((const uint8_t *)W, (int)N, nblk, (const half_t *)X, (int)T, (int)K,
                                                              (const half_t *)bias, (half_t *)Y, g_gemm_big >> 4);
        }
        return IFA_OK;
    }
    if (T > 128 && ifa_cdiv(N, GEMM_ROWS) * ifa_cdiv(T, 128) >= 256) {
        // enough tiles to fill the chip twice over with 128-token tiles: the dequantised block is reused for 4 MFMA tiles
        constexpr int MT4 = 4;
        dim3 grid(ifa_cdiv(N, GEMM_ROWS), ifa_cdiv(T, 32 * MT4));
        k_gemm_q<DT, MT4, false><<<grid, dim3(GEMM_THREADS), 2 * slab, s>>>((const uint8_t *)W, (int)N, nblk, (const half_t *)X, (int)T,
                                                                           (int)K, (const half_t *)bias, (half_t *)Y);
        return IFA_OK;
    }
    if (T <= 32) {       // one 32-token tile: half the activation staging of the 64-token variant
        constexpr int MT1 = 1, NW = (CAP <= 32 && DT != F16) ? 8 : 4;   // 8 waves share a tile's K range (64-value blocks: 4, the stages would spill): 128 tiles of a 4096-row matrix leave half the CUs idle,
        dim3 grid(ifa_cdiv(N, 32), 1);      // the serial length of a wave (dequantisation VALU + load waits) is what counts
        const size_t smem = std::max(NW * (slab / 2), (size_t)(NW - 1) * MT1 * 16 * 64 * 4);
        k_gemm_q<DT, MT1, true, NW><<<grid, dim3(NW * 64), smem, s>>>((const uint8_t *)W, (int)N, nblk, (const half_t *)X, (int)T,
                                                                      (int)K, (const half_t *)bias, (half_t *)Y);
        return IFA_OK;
    }
    if (T <= 128) {      // weight-stream bound: 32-row tiles, K split over the 4 waves
        dim3 grid(ifa_cdiv(N, 32), ifa_cdiv(T, 32 * MT));
        constexpr int NW = 4;               // (8 waves x 64-token tiles need more than 256 registers per lane)
        const size_t smem = std::max(NW * slab, (size_t)(NW - 1) * MT * 16 * 64 * 4);
        k_gemm_q<DT, MT, true, NW><<<grid, dim3(NW * 64), smem, s>>>((const uint8_t *)W, (int)N, nblk, (const half_t *)X, (int)T,
                                                                     (int)K, (const half_t *)bias, (half_t *)Y);
    } else {
        dim3 grid(ifa_cdiv(N, GEMM_ROWS), ifa_cdiv(T, 32 * MT));
        k_gemm_q<DT, MT, false><<<grid, dim3(GEMM_THREADS), slab, s>>>((const uint8_t *)W, (int)N, nblk, (const half_t *)X, (int)T,
                                                                      (int)K, (const half_t *)bias, (half_t *)Y);
    }
    return IFA_OK;
}

// Grouped product of a mixture-of-experts step (ifa_moe.h): every tile of the device-side table is <= 64 rows of one
// expert; weight-stream regime (the split-K kernel of the T <= 128 range), max_tiles = upper bound of the table's length
namespace ifa {
int gemm_q_grouped(int w_dtype, const MoeGroup &grp, size_t N, size_t K, const void *X, void *Y, int max_tiles, int tile_rows, hipStream_t s)
{
    const int cap = w_dtype == F16 ? 32 : block_capacity(w_dtype);
    if (cap <= 1 || K % (size_t)cap != 0 || K % 8 != 0) return ifa_fail(IFA_ERR_ARG, "grouped GEMM: dtype %d / cols %zu", w_dtype, K);
    if (tile_rows != 64 && tile_rows != 128) return ifa_fail(IFA_ERR_ARG, "grouped GEMM: tile of %d rows", tile_rows);
    if (max_tiles <= 0) return IFA_OK;
    auto launch = [&](auto tag) {
        constexpr int DT = decltype(tag)::value;
        constexpr int CAP = (DT == F16) ? 32 : block_capacity(DT);
        const size_t slab = (size_t)32 * 2 * (2 * CAP * 2 + 16);
        if (tile_rows == 64) {          // few rows per expert: weight-stream regime, 32-row tiles with K split over 4 waves
            constexpr int MT = 2, NW = 4;
            const size_t smem = std::max(NW * slab, (size_t)(NW - 1) * MT * 16 * 64 * 4);
            dim3 grid(ifa_cdiv(N, 32), (unsigned)max_tiles);
            k_gemm_q<DT, MT, true, NW><<<grid, dim3(NW * 64), smem, s>>>(nullptr, (int)N, (int)(K / CAP), (const half_t *)X, 0, (int)K, nullptr,
                                                                         (half_t *)Y, grp);
        } else {                        // long prompts: 128 rows x 128 weight rows per workgroup, each decoded block feeds 4 MFMA tiles
            constexpr int MT4 = 4;
            dim3 grid(ifa_cdiv(N, GEMM_ROWS), (unsigned)max_tiles);
            k_gemm_q<DT, MT4, false><<<grid, dim3(GEMM_THREADS), 2 * slab, s>>>(nullptr, (int)N, (int)(K / CAP), (const half_t *)X, 0, (int)K, nullptr,
                                                                               (half_t *)Y, grp);
        }
    };
    if (w_dtype == F16) launch(std::integral_constant<int, F16>());
    else { IFA_DISPATCH_QUANT_DTYPE(w_dtype, launch(std::integral_constant<int, DT>())); }
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}
}

extern "C" int ifa_gemm_big_tiles(int on)
{
    const int prev = g_gemm_big;
    if (on >= 0) g_gemm_big = on;      // bit 0: on; bits 4..: ablation switches of k_gemm_big (measurement)
    return prev;
}

extern "C" int ifa_gemm(int w_dtype, const void *W, size_t rows, size_t cols, const void *x_f16, size_t tokens,
                        const void *bias_f16, void *y_f16, ifa_stream stream)
{
    IFA_REQUIRE(W && x_f16 && y_f16, "ifa_gemm: null pointer");
    if (rows == 0 || tokens == 0) return IFA_OK;
    const int cap = w_dtype == F16 ? 32 : block_capacity(w_dtype);
    IFA_REQUIRE(cap > 1, "ifa_gemm: unsupported weight dtype %d", w_dtype);
    IFA_REQUIRE(cols > 0 && cols % (size_t)cap == 0 && cols % 8 == 0, "ifa_gemm: cols %zu must be a multiple of %d", cols, cap);
    IFA_REQUIRE(rows < (1u << 30) && cols < (1u << 30) && tokens <= 65535u * 64u, "ifa_gemm: shape too large");
    hipStream_t s = ifa_s(stream);
    // MFMA-bound sizes: dequantise once + the library's F16 GEMM (ifa_gemm_lt.hip); anything it declines runs below
    if (gemm_lt_wanted(tokens) && gemm_lt(w_dtype, W, rows, cols, x_f16, tokens, bias_f16, y_f16, s) == IFA_OK) {
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    if (w_dtype == F16) {
        launch_gemm<F16>(W, rows, cols, x_f16, tokens, bias_f16, y_f16, s);
    } else {
        IFA_DISPATCH_QUANT_DTYPE(w_dtype, launch_gemm<DT>(W, rows, cols, x_f16, tokens, bias_f16, y_f16, s));
    }
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}
