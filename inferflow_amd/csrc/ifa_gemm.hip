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
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include "ifa_host.h"
#include "ifa_codec.h"
#include "ifa_moe.h"
#include "ifa_math.h"
#include "ifa_gemm_rows_mfma.h"
#include "ifa_gemm_big.h"

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
        blk.load_global(W + (row * (size_t)nblk + (size_t)b) * BB);
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
        // (weights are device memory: global loads whatever the pointer's origin -- the MoE expert table hands it over through
        //  memory, and FLAT loads would count on lgkmcnt next to the LDS traffic of the step loop)
        const uint8_t *p = W + (row * (size_t)nblk + (size_t)b) * BB;
        typedef const __attribute__((address_space(1))) u32x4 gu4;
        typedef const __attribute__((address_space(1))) uint32_t gu1;
        if constexpr (DT == F16) {
#pragma unroll
            for (int i = 0; i < CAP / 8; i++) f16v[i] = ((gu4 *)p)[i];
        } else if constexpr (Q4FAST) {
#pragma unroll
            for (int i = 0; i < 5; i++) q4[i] = ((gu1 *)p)[i];
        } else {
            blk.load_global(p);
        }
    }
    // materialise the raw registers here (an empty asm the compiler cannot move a use across)
    __device__ __forceinline__ void pin()
    {
        if constexpr (DT == F16) {
#pragma unroll
            for (int i = 0; i < CAP / 8; i++) asm volatile("" : "+v"(f16v[i]));
        } else if constexpr (Q4FAST) {
#pragma unroll
            for (int i = 0; i < 5; i++) asm volatile("" : "+v"(q4[i]));
        }
    }
    // 8 consecutive values (chunk m of the block) -- F16 and the 20-byte Q4 blocks only
    __device__ __forceinline__ half8_t chunk(int m) const
    {
        if constexpr (DT == F16) return __builtin_bit_cast(half8_t, f16v[m]);
        else {
            static_assert(Q4FAST || DT == F16, "chunk decoder");
            const float base = hbits2f((uint16_t)(q4[0] & 0xFFFFu)), scale = hbits2f((uint16_t)(q4[0] >> 16));
            const uint32_t lo = q4[1 + m] & 0x0F0F0F0Fu, hi = (q4[1 + m] >> 4) & 0x0F0F0F0Fu;
            const float ql[4] = {ubyte_f32<0>(lo), ubyte_f32<1>(lo), ubyte_f32<2>(lo), ubyte_f32<3>(lo)};
            const float qh[4] = {ubyte_f32<0>(hi), ubyte_f32<1>(hi), ubyte_f32<2>(hi), ubyte_f32<3>(hi)};
            half8_t h;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                h[2 * b] = f2h(__builtin_fmaf(ql[b], scale, base));
                h[2 * b + 1] = f2h(__builtin_fmaf(qh[b], scale, base));
            }
            return h;
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
// Large T (prefill, MFMA-bound): BM tokens x BN weight rows per workgroup (256 x 256, 128 x 256 or 128 x 128 by the
// number of tiles the problem offers), K walked in steps of 64, every wave multiplies AND stages.
//  * the weight tile is dequantised ONCE per workgroup and step into LDS (values rounded to half exactly like the
//    reference's Dequantize tensor) and shared by the waves -- the small-T kernel above decodes a block per lane and
//    MFMA tile, which bounds it at ~0.5 PFLOP/s (VALU per MFMA).  One quant block per thread and step: its raw bytes are
//    requested a step ahead, decoded between the MFMA groups of the current step (VALU in the shadow of the matrix pipe)
//    and written into the other LDS buffer;
//  * the activation tile goes HBM / L2 -> LDS directly (global_load_lds_dwordx4, no registers): a wave writes 8 rows x
//    128 bytes per instruction, a step ahead into the other buffer;
//  * both LDS images are [rows][64 halfs] with the eight 16-byte chunks of row r at slot c ^ ((r >> 1) & 7): the
//    fragment reads of 16 consecutive lanes (16 rows, one chunk) cover all 64 banks once.  The direct loads write LDS
//    linearly, so the permutation is applied to their SOURCE addresses (cdna_hip_programming.md 5.4 rule 21);
//  * workgroup ids are dealt to the 8 XCDs round-robin: the id is remapped so that an XCD works on a compact block of
//    tiles (bands of 1024 tokens x a few weight tiles) whose operand tiles its L2 shares;
//  * the step's barrier sits before the last MFMA group of the step (see the K loop), the epilogue rounds the products
//    to F16, parks them in an LDS tile and writes 16-byte pieces of output rows (residual / GLU applied on the way out).
// Same arithmetic as k_gemm_q: fp32 accumulation of half products in ascending 16-column groups, one F16 rounding, bias as a half add.
constexpr int PF_BK = 64;
// weight-tile row of the p-th (row, block) slot: 8 consecutive lanes (4 slots x 2 blocks: one ds_write_b128 group) take
// rows 0, 2, 4, 6 (then 1, 3, 5, 7) of a group of 8 -- four different swizzle values, 8 distinct 16-byte bank groups
__device__ __forceinline__ int pf_row_of(int p) { return (p & ~7) | ((p & 3) << 1) | ((p >> 2) & 1); }
typedef __attribute__((address_space(3))) void pf_lds_t;
typedef const __attribute__((address_space(1))) void pf_glb_t;

// Launch arguments: GmArgs (ifa_gemm_rows_mfma.h -- up to three matrices as one row space, per-set or virtual-row
// outputs, bias, residual, GLU pair; W[] / W1 are REFERENCE-layout rows here) + the tile geometry.
// EPI: GM_PLAIN | GM_RESIDUAL (Y = half(res + y), TensorOpr::Add) | GM_GLU (a weight tile = BN / 2 rows of w1 and the
// same BN / 2 rows of w3, both halves meet in the epilogue's LDS tile: Y = half(half(act(y1)) * y3)).
// KS = 2 (round 4, 128 x 128 tiles of a product that offers no more tiles than CUs: wo and w2 of a 1024-token prompt are 256 tiles of
// four waves -- one workgroup per CU, the matrix pipe 25 % busy): two workgroups per tile, each walks half of K; the first half's
// fp32 accumulators go through memory (write-through stores, one flag per tile) to the workgroup of the second half, which adds them
// IN THAT ORDER (first half + second half: deterministic) and runs the epilogue.  The grid lists all first halves, then all second
// halves: a second half is never resident before its first half.  part / flags: per-stream scratch of the launcher.
// (BigGeo: ifa_gemm_big.h)

template <int DT, int BM, int BN, int WM, int WN, int EPI = GM_PLAIN, int BK = PF_BK, int KS = 1>
__global__ void __launch_bounds__(WM * WN * 64) k_gemm_big(const GmArgs P, const BigGeo G)
{
    static_assert(KS == 0 || KS == 1 || KS == 2 || KS == 4, "split-K: one, two or four workgroups per tile; 0: the stream-K schedule");
    constexpr bool SK = KS == 0;                        // stream-K (see below the tile order)
    constexpr int KSD = SK ? 1 : KS;
    // BK: columns per K step -- 64, or 128 for 128 x 128 tiles that run one workgroup per CU (half the barriers per product)
    constexpr int ROWB = BK * 2, CPR = BK / 8, RPP = 1024 / ROWB;     // LDS row bytes, 16-byte chunks per row, rows per direct-to-LDS piece
    auto swz = [](int r) { return BK == 64 ? ((r >> 1) & 7) : (r & 15); };
    constexpr bool GLU = EPI == GM_GLU;
    constexpr int BNE = GLU ? BN / 2 : BN;              // rows of ONE matrix per weight tile
    const int T = P.T, K = G.K, nblk = P.nblk, tiles_m = G.tiles_m;
    const half_t *__restrict__ X = P.X;
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int CAP = (DT == F16) ? 32 : block_capacity(DT);
    constexpr int BPS = BK / CAP;                       // quant blocks per row and step
    constexpr int CPB = CAP / 8;                        // 16-byte chunks of halfs per block
    constexpr int NB = BN * BPS;                        // blocks of the weight tile per step
    constexpr int WB = (NB + NT - 1) / NT;              // blocks a thread dequantises per step
    constexpr int TA = BM / WM / 32, TB = BN / WN / 32; // 32 x 32 accumulator tiles per wave
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int NA = 2;                               // LDS: NA activation buffers, then 2 weight buffers
    constexpr int AI = BM / RPP / NW;                   // direct-to-LDS instructions per wave and step (1 KB = RPP rows each)
    static_assert(BPS >= 1 && BM % (2 * RPP * NW) == 0 && (BK == 64 || BK == 128), "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    // Stream-K schedule (KS = 0; round 6, 256 x 256 tiles of a product whose tile count is no multiple of the CU count: w1 | w3 of a
    // 1024-token prompt are 344 tiles on 256 CUs, wq | wk | wv 192): ONE workgroup per CU; each first takes G.sk_full whole tiles,
    // then an equal share of the K steps of the G.sk_rem tiles that are left -- steps [b(w), b(w + 1)) of the remainder's linear
    // (tile, step) space, b(w) = w * sk_rem * S / grid -- which touches at most two tiles.  A share that does not END its tile leaves
    // its fp32 sums in the tile's scratch slot j (j = how many shares precede it in the tile) and bumps the tile's counter; the share
    // that ends the tile waits for the j others, adds them in ONE order (own + slot 0 + slot 1 + ...: deterministic) and runs the
    // epilogue.  A workgroup with two shares takes the one that STARTS a tile first: no workgroup waits before its partial sums are
    // out, so there is no chain of waits through the grid.
    const int bid = (int)blockIdx.x;
    int nseg = 1, sk_b0 = 0, sk_b1 = 0, sk_tA = 0, sk_S = 0, sk_RS = 0;
    bool sk_two = false;
    if constexpr (SK) {
        sk_S = K / BK; sk_RS = G.sk_rem * sk_S;
        sk_b0 = (int)((long long)bid * sk_RS / (int)gridDim.x); sk_b1 = (int)((long long)(bid + 1) * sk_RS / (int)gridDim.x);
        sk_tA = sk_b0 / sk_S;
        sk_two = sk_b1 > (sk_tA + 1) * sk_S;
        nseg = G.sk_full + (sk_b1 > sk_b0 ? 1 : 0) + (sk_two ? 1 : 0);
    }
#pragma unroll 1
    for (int seg = 0; seg < nseg; seg++) {
    // (stream-K: every lane constant below is derived from a laundered thread id, so that the compiler does not hoist the constants of
    //  the fix-up and the epilogue out of this loop and keep them alive through the K loop, where they would spill)
    int tid_l = tid0;
    if constexpr (SK) asm volatile("" : "+v"(tid_l));
    const int tid = tid_l, lane = tid & 63;
    // XCD-aware tile order (bijective for any grid size)
    int wg, kbeg = 0, nsteps = K / BK / KSD, sk_kind = 0, sk_j = 0, sk_rt = 0;       // sk_kind: 0 whole tile, 1 share that leaves partial sums, 2 share that ends the tile
    if constexpr (SK) {
        if (seg < G.sk_full) {
            const int nwg = G.sk_full * (int)gridDim.x, orig = seg * (int)gridDim.x + bid, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
            wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        } else {
            if (sk_two && seg == G.sk_full) { sk_rt = sk_tA + 1; kbeg = 0; nsteps = sk_b1 - sk_rt * sk_S; sk_kind = 1; sk_j = 0; }
            else {
                sk_rt = sk_tA; kbeg = sk_b0 - sk_tA * sk_S;
                const int kend = min(sk_b1, (sk_tA + 1) * sk_S) - sk_tA * sk_S;
                nsteps = kend - kbeg; sk_kind = kend == sk_S ? 2 : 1;
                while (bid - sk_j >= 1 && (int)((long long)(bid - sk_j) * sk_RS / (int)gridDim.x) > sk_tA * sk_S) sk_j++;
            }
            wg = G.sk_full * (int)gridDim.x + sk_rt;
        }
    } else {
        const int nwg = (int)gridDim.x / KSD, orig = bid % nwg, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int kz = KSD == 1 ? 0 : bid / ((int)gridDim.x / KSD);       // which half of K (wave-uniform)
    // tiles in bands of GM token tiles, weight tiles next, token tiles of the band fastest: the workgroups an XCD runs
    // at the same time form a compact block (GM token tiles x a few weight tiles) whose operand tiles its L2 shares
    constexpr int GM = 1024 / BM;
    const int tiles_n = SK ? G.sk_tiles_n : (int)gridDim.x / KSD / tiles_m;
    const int band = wg / (GM * tiles_n), within = wg % (GM * tiles_n), band_m = min(GM, tiles_m - band * GM);
    const int t0 = (band * GM + within % band_m) * BM, tn = G.tn0 + within / band_m;
    // weight tile -> (matrix, first row); everything selected by VALUE from the argument block (no indexed struct access)
    const int set = (tn >= G.tile0[1] ? 1 : 0) + (tn >= G.tile0[2] ? 1 : 0);
    const int n0 = (tn - (set == 0 ? 0 : (set == 1 ? G.tile0[1] : G.tile0[2]))) * BNE;
    const int N = set == 0 ? P.rows[0] : (set == 1 ? P.rows[1] : P.rows[2]);
    const uint8_t *__restrict__ W = set == 0 ? P.W[0] : (set == 1 ? P.W[1] : P.W[2]);
    const half_t *__restrict__ bias = set == 0 ? P.bias[0] : (set == 1 ? P.bias[1] : P.bias[2]);
    const int s0 = SK ? kbeg : kz * nsteps;                 // this workgroup's K steps: [s0, s0 + nsteps)
    // ---- activation tile: per-lane source pointers of this wave's 8-row pieces
    const half_t *xsrc[AI];
#pragma unroll
    for (int j = 0; j < AI; j++) {
        const int row = (wave * AI + j) * RPP + lane / CPR;
        const int c = (lane % CPR) ^ swz(row);
        xsrc[j] = X + (size_t)min(t0 + row, T - 1) * P.ldx + c * 8;
    }
    auto stage_x = [&](int step, int buf, int j0, int j1) {
#pragma unroll
        for (int j = j0; j < j1; j++)
            __builtin_amdgcn_global_load_lds((pf_glb_t *)(xsrc[j] + (size_t)(s0 + step) * BK),
                                             (pf_lds_t *)(smem + (size_t)buf * A_BYTES + (size_t)(wave * AI + j) * 1024), 16, 0, 0);
    };
    // ---- weight tile: (row, block) slots over the threads.  BK = 64: two blocks of a row on adjacent lanes, rows in the
    // order of pf_row_of; BK = 128: consecutive lanes take consecutive rows of ONE block column (8 lanes of a
    // ds_write_b128 group: 8 different swizzle values) -- both orders are conflict-free on the stores
    auto slot_row = [](int idx) { return BK == 64 ? pf_row_of(idx / BPS) : idx % BN; };
    auto slot_blk = [](int idx) { return BK == 64 ? idx % BPS : idx / BN; };
    WRaw<DT, CAP> wr[WB];
    auto fetch_w = [&](int step) {
#pragma unroll
        for (int j = 0; j < WB; j++) {
            const int idx = min(tid + j * NT, NB - 1);
            const int nl = slot_row(idx), bb = slot_blk(idx);
            if constexpr (GLU) wr[j].load(nl < BNE ? W : P.W1, (size_t)min(n0 + (nl < BNE ? nl : nl - BNE), N - 1), nblk, (s0 + step) * BPS + bb);
            else wr[j].load(W, (size_t)min(n0 + nl, N - 1), nblk, (s0 + step) * BPS + bb);
        }
    };
    // the raw bytes of the block(s) being dequantised this step (wr is refilled for the step after next meanwhile);
    // formats without a per-chunk decoder are decoded whole at the head of the step
    constexpr bool CHUNKED = WRaw<DT, CAP>::Q4FAST || DT == F16;
    WRaw<DT, CAP> wc[WB];
    half_t wv[CHUNKED ? 1 : WB][CHUNKED ? 8 : CAP];
    auto take_w = [&]() {
#pragma unroll
        for (int j = 0; j < WB; j++) {
            if constexpr (CHUNKED) { wc[j] = wr[j]; wc[j].pin(); }
            else wr[j].decode(true, wv[j]);
        }
    };
    auto store_w = [&](int buf, int m0, int m1) {           // chunks [m0, m1) of every block of this thread
        char *Bs = smem + (size_t)NA * A_BYTES + (size_t)buf * B_BYTES;
#pragma unroll
        for (int j = 0; j < WB; j++) {
            const int idx = tid + j * NT;
            if (NB % NT != 0 && idx >= NB) continue;
            const int nl = slot_row(idx), b = slot_blk(idx);
#pragma unroll
            for (int m = m0; m < m1; m++) {
                half8_t h;
                if constexpr (CHUNKED) h = wc[j].chunk(m);
                else {
#pragma unroll
                    for (int e = 0; e < 8; e++) h[e] = wv[j][8 * m + e];
                }
                *reinterpret_cast<half8_t *>(Bs + (size_t)nl * ROWB + (size_t)(((b * CPB + m) ^ swz(nl)) << 4)) = h;
            }
        }
    };
    // ---- fragments: lane (i, g) reads row i of a 32-row tile, chunk 2 * ks + g
    const int i = lane & 31, g = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int lc = g ^ swz(i);                          // (2 ks + g) ^ swizzle(i) == (2 ks) ^ lc  (tile rows start at multiples of 32)
    const int a_off = (wm * (BM / WM) + i) * ROWB, b_off = NA * A_BYTES + (wn * (BN / WN) + i) * ROWB;
    f32x16_t acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; a++)
#pragma unroll
        for (int b = 0; b < TB; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    // K loop.  A step = four MFMA groups (16 columns each) of buffer `cur`; the fragments of a group are requested ahead of
    // the products of the group before it, a share of the next step's weight chunks is dequantised and stored behind the
    // first three groups.  The step's barrier sits BEFORE the last group: by then every wave has read its last fragments of
    // `cur` and finished its part of the next tile, so the first fragments of the next buffer are requested right behind
    // the barrier and land under the last group's products -- no LDS round trip is exposed at a step boundary.
    half8_t af[2][TA], bf[2][TB];
    auto frags = [&](int buf, int ks, half8_t (&fa)[TA], half8_t (&fb)[TB]) {
        const char *As = smem + (size_t)buf * A_BYTES, *Bs = smem + (size_t)buf * B_BYTES;
        const int so = ((2 * ks) ^ lc) << 4;
#pragma unroll
        for (int a = 0; a < TA; a++) fa[a] = *reinterpret_cast<const half8_t *>(As + a_off + a * 32 * ROWB + so);
#pragma unroll
        for (int b = 0; b < TB; b++) fb[b] = *reinterpret_cast<const half8_t *>(Bs + b_off + b * 32 * ROWB + so);
    };
    auto mma = [&](const half8_t (&fa)[TA], const half8_t (&fb)[TB]) {
#pragma unroll
        for (int a = 0; a < TA; a++)
#pragma unroll
            for (int b = 0; b < TB; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    };

    stage_x(0, 0, 0, AI);
    fetch_w(0);
    take_w();
    store_w(0, 0, CPB);
    fetch_w(min(1, nsteps - 1));
    __syncthreads();
    frags(0, 0, af[0], bf[0]);
    constexpr int NKS = BK / 16;                         // MFMA groups per step (even: the fragment sets alternate cleanly)
    for (int step = 0; step + 1 < nsteps; step++) {
        const int cur = step & 1;
        // the copy is pinned AHEAD of the direct-to-LDS loads: with one of those in flight the compiler waits vmcnt(0) at
        // the next use of an ordinary load's result, which would expose the whole latency of the tile just requested
        take_w();                                        // (the barrier of the previous step waited for its bytes)
        stage_x(step + 1, cur ^ 1, 0, AI);
        fetch_w(min(step + 2, nsteps - 1));              // a whole step to land; the last one is a harmless repeat
#pragma unroll
        for (int ks = 0; ks + 1 < NKS; ks++) {
            frags(cur, ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
            mma(af[ks & 1], bf[ks & 1]);
            store_w(cur ^ 1, (ks * CPB + NKS - 2) / (NKS - 1), ((ks + 1) * CPB + NKS - 2) / (NKS - 1));
        }
        __syncthreads();
        frags(cur ^ 1, 0, af[0], bf[0]); mma(af[1], bf[1]);
    }
    {
        const int cur = (nsteps - 1) & 1;
#pragma unroll
        for (int ks = 0; ks + 1 < NKS; ks++) {
            frags(cur, ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
            mma(af[ks & 1], bf[ks & 1]);
        }
        mma(af[1], bf[1]);
    }
    if constexpr (KS > 1) {
        // thread tid's accumulators as 8-byte words q * NT + tid of the tile's scratch slot kz (coalesced either way); the workgroup of
        // the LAST part of K waits for the KS - 1 others (one counter per tile) and adds their sums in K order: part 0 + part 1 + ... + own
        constexpr int NQ = TA * TB * 8;
        unsigned long long *pt = G.part + (size_t)wg * ((size_t)(KS - 1) * NQ * NT);
        unsigned *flag = G.flags + wg;
        if (kz < KS - 1) {
            if (G.dbg_skip) return;                  // (tests: a partner that never publishes)
            unsigned long long *mine = pt + (size_t)kz * NQ * NT;
#pragma unroll
            for (int a = 0; a < TA; a++)
#pragma unroll
                for (int b = 0; b < TB; b++)
#pragma unroll
                    for (int rp = 0; rp < 8; rp++) {
                        const float f0 = acc[a][b][2 * rp], f1 = acc[a][b][2 * rp + 1];       // (scalars first: a bit_cast of a vector ELEMENT read element 0 every time)
                        const unsigned long long v = (unsigned long long)__builtin_bit_cast(uint32_t, f0) | ((unsigned long long)__builtin_bit_cast(uint32_t, f1) << 32);
                        __hip_atomic_store(mine + (size_t)((a * TB + b) * 8 + rp) * NT + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the partial sums are in memory ...
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... before the counter says so
            return;
        }
        if (tid == 0) {
            const long long t_wait = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(KS - 1)) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t_wait > WAIT_TIMEOUT_TICKS) {      // the first halves are not resident: leave a code and go on (garbage sums; the host fails the
                    if (G.err) __hip_atomic_store(G.err, 0x81u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // call and switches the waiting launches off)
                    break;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < TA; a++)
#pragma unroll
            for (int b = 0; b < TB; b++)
#pragma unroll
                for (int rp = 0; rp < 8; rp++) {
                    float s0f = 0.0f, s1f = 0.0f;
#pragma unroll
                    for (int z = 0; z < KS - 1; z++) {
                        const unsigned long long v = __hip_atomic_load(pt + (size_t)z * NQ * NT + (size_t)((a * TB + b) * 8 + rp) * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float p0 = __builtin_bit_cast(float, (uint32_t)v), p1 = __builtin_bit_cast(float, (uint32_t)(v >> 32));
                        s0f = z == 0 ? p0 : s0f + p0; s1f = z == 0 ? p1 : s1f + p1;
                    }
                    acc[a][b][2 * rp] = s0f + acc[a][b][2 * rp];
                    acc[a][b][2 * rp + 1] = s1f + acc[a][b][2 * rp + 1];
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // ready for the next launch on this stream
    }
    int etid = tid;
    if constexpr (SK) asm volatile("" : "+v"(etid));      // (and once more behind the K loop)
    const int ei = etid & 31, eg = (etid & 63) >> 5;
    bool do_epi = true;
    if constexpr (SK) {
        constexpr int NQ = TA * TB * 8;
        unsigned long long *pt = G.part + (size_t)sk_rt * ((size_t)G.sk_slots * NQ * NT);
        unsigned *flag = G.flags + sk_rt;
        if (sk_kind == 1) {
            unsigned long long *mine = pt + (size_t)sk_j * NQ * NT;
#pragma unroll
            for (int a = 0; a < TA; a++)
#pragma unroll
                for (int b = 0; b < TB; b++)
#pragma unroll
                    for (int rp = 0; rp < 8; rp++) {
                        const float f0 = acc[a][b][2 * rp], f1 = acc[a][b][2 * rp + 1];
                        const unsigned long long v = (unsigned long long)__builtin_bit_cast(uint32_t, f0) | ((unsigned long long)__builtin_bit_cast(uint32_t, f1) << 32);
                        __hip_atomic_store(mine + (size_t)((a * TB + b) * 8 + rp) * NT + etid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the partial sums are in memory ...
            __syncthreads();
            if (etid == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... before the counter says so
            do_epi = false;
        }
        // (the sums of the other shares are added in a loop whose trip count is zero for every other kind of share: a conditional block
        //  around the additions made the compiler keep two copies of the 128 accumulators alive -- 50 to 100 spilled registers)
        const int nz = sk_kind == 2 ? sk_j : 0;
        if (nz > 0) {
            if (etid == 0) {
                const long long t_wait = wall_clock64();
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)sk_j) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t_wait > WAIT_TIMEOUT_TICKS) {      // (a partner is not resident: leave a code and go on -- the host fails the call)
                        if (G.err) __hip_atomic_store(G.err, 0x82u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll 1
        for (int z = 0; z < nz; z++) {                                     // own + slot 0 + slot 1 + ...: one fixed order
            const unsigned long long *pz = pt + (size_t)z * NQ * NT + etid;
#pragma unroll
            for (int a = 0; a < TA; a++)
#pragma unroll
                for (int b = 0; b < TB; b++)
#pragma unroll
                    for (int rp = 0; rp < 8; rp++) {
                        const unsigned long long v = __hip_atomic_load(pz + (size_t)((a * TB + b) * 8 + rp) * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        acc[a][b][2 * rp] += __builtin_bit_cast(float, (uint32_t)v);
                        acc[a][b][2 * rp + 1] += __builtin_bit_cast(float, (uint32_t)(v >> 32));
                    }
        }
        if (nz > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (etid == 0) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // ready for the next launch on this stream
        }
    }
    if (do_epi) {
    // ---- epilogue
    half_t *__restrict__ Yo; int ldo, vrow0 = 0;
    if (P.Yset[0]) { Yo = set == 0 ? P.Yset[0] : (set == 1 ? P.Yset[1] : P.Yset[2]); ldo = set == 0 ? P.ldyset[0] : (set == 1 ? P.ldyset[1] : P.ldyset[2]); }
    else { Yo = P.Y; ldo = P.ldy; vrow0 = (set >= 1 ? P.rows[0] : 0) + (set >= 2 ? P.rows[1] : 0); }
    // products -> F16 (+ bias) -> an LDS tile [BM][BN] -> 16-byte pieces of output rows (residual / GLU applied on the way out).
    // A lane holds ONE output column (32 lanes: 32 adjacent columns of a token row); adjacent lanes trade every other value
    // (DPP quad swap) so that each writes a (column, column + 1) pair as one dword; rows padded by 64 bytes: the even / odd
    // lanes of a write land in two different token rows, 16 banks apart.
    constexpr int WCOLS = BN / WN;                      // weight rows (output columns) of a wave
    constexpr int CROW = BN * 2 + 64;
    __syncthreads();                                    // every wave is done with the operand tiles
    {
        const bool odd = ei & 1;
#pragma unroll
        for (int b = 0; b < TB; b++) {
            const int cl = wn * WCOLS + b * 32 + ei;
            float bv = 0.0f; bool hb = false;
            if constexpr (GLU) {
                const half_t *bp = cl < BNE ? bias : P.bias1;
                const int n = n0 + (cl < BNE ? cl : cl - BNE);
                if (bp) { hb = true; bv = h2f(bp[min(n, N - 1)]); }
            } else if (bias) { hb = true; bv = h2f(bias[min(n0 + cl, N - 1)]); }
#pragma unroll
            for (int a = 0; a < TA; a++)
#pragma unroll
                for (int rp = 0; rp < 8; rp++) {
                    half_t y0 = f2h(acc[a][b][2 * rp]), y1 = f2h(acc[a][b][2 * rp + 1]);
                    if (hb) { y0 = f2h(h2f(y0) + bv); y1 = f2h(h2f(y1) + bv); }
                    const uint32_t u0 = __builtin_bit_cast(uint16_t, y0), u1 = __builtin_bit_cast(uint16_t, y1);
                    const uint32_t send = odd ? u0 : u1;
                    const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, false);   // quad_perm [1, 0, 3, 2]
                    const uint32_t packed = odd ? (recv | (u1 << 16)) : (u0 | (recv << 16));
                    const int rr = 2 * rp + (odd ? 1 : 0);
                    const int tl = wm * (BM / WM) + a * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * eg;
                    *reinterpret_cast<uint32_t *>(smem + (size_t)tl * CROW + (size_t)(cl & ~1) * 2) = packed;
                }
        }
    }
    __syncthreads();
    {
        constexpr int VEC = BNE / 8;                    // 16-byte pieces per output row of the tile
        for (int idx = etid; idx < BM * VEC; idx += NT) {
            const int tl = idx / VEC, v = idx % VEC, tok = t0 + tl, n = n0 + v * 8;
            if (tok >= T || n >= N) continue;
            half8_t y = *reinterpret_cast<const half8_t *>(smem + (size_t)tl * CROW + (size_t)v * 16);
            if constexpr (GLU) {
                const half8_t u = *reinterpret_cast<const half8_t *>(smem + (size_t)tl * CROW + (size_t)(BNE + v * 8) * 2);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const half_t act = f2h(act_fn(h2f(y[e]), P.act_kind));                 // TensorOpr::Activation -> F16
                    y[e] = f2h(h2f(act) * h2f(u[e]));                                      // TensorOpr::Mul
                }
            } else if constexpr (EPI == GM_RESIDUAL) {
                const half8_t r = *reinterpret_cast<const half8_t *>(P.res + (size_t)tok * P.ldres + vrow0 + n);
#pragma unroll
                for (int e = 0; e < 8; e++) y[e] = f2h(h2f(r[e]) + h2f(y[e]));             // TensorOpr::Add
            }
            *reinterpret_cast<half8_t *>(Yo + (size_t)tok * ldo + vrow0 + n) = y;
        }
    }
    }      // do_epi
    if (seg + 1 < nseg) __syncthreads();      // (the next share stages into the LDS the epilogue has just read)
    }      // shares of this workgroup
}

} // namespace ifa

using namespace ifa;

namespace ifa { void attn_release_stream(int dev, hipStream_t s); }      // ifa_attn.hip: the score-tile workspace of the stream
namespace ifa { void rows_kparts_release(int dev, hipStream_t s); }      // ifa_gemm_rows_mo.hip: the K-parts scratch of the stream

static int gemm_num_cus()
{
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t prop; n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
    return n;
}
static int g_gemm_big = 1;       // ifa_gemm_big_tiles(0 / 1): the large-tile kernel for T > 128 (default on; blocks of <= 32 values -- the 64-value formats keep k_gemm_q); bits 8-9: force a tile shape (measurement)

// tile shape by estimated time: rounds of workgroups over the chip x the measured cost of one round
// per-(device, stream) scratch of the split-K launches: the first halves' accumulators + one flag per tile (flags zero between launches)
constexpr size_t SPLITK_FLAG_BYTES = 16384;      // one counter per tile, in front of the partial sums
struct SplitKScratch { void *p = nullptr; size_t bytes = 0; };
static std::mutex g_splitk_mu;
static std::map<std::pair<int, hipStream_t>, SplitKScratch> g_splitk;
namespace ifa { int gemm_splitk_scratch(hipStream_t s, size_t part_bytes, size_t tiles, void **out); }
int ifa::gemm_splitk_scratch(hipStream_t s, size_t part_bytes, size_t tiles, void **out)
{
    int dev = 0;
    IFA_HIP_CHECK(hipGetDevice(&dev));
    if (tiles * 4 > SPLITK_FLAG_BYTES) return ifa_fail(IFA_ERR_ARG, "split-K: %zu tiles", tiles);
    const size_t need = SPLITK_FLAG_BYTES + part_bytes;
    std::lock_guard<std::mutex> lk(g_splitk_mu);
    SplitKScratch &sc = g_splitk[std::make_pair(dev, s)];
    if (sc.bytes < need) {
        IFA_HIP_CHECK(hipStreamSynchronize(s));
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr; sc.bytes = 0;
        IFA_HIP_CHECK(hipMalloc(&sc.p, need));
        IFA_HIP_CHECK(hipMemsetAsync(sc.p, 0, need, s));          // (flags: zero; the launches that follow on this stream see it)
        sc.bytes = need;
    }
    *out = sc.p;
    return IFA_OK;
}
static void gemm_splitk_release(int dev, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_splitk_mu);
    auto it = g_splitk.find(std::make_pair(dev, s));
    if (it == g_splitk.end()) return;
    if (it->second.p) (void)hipFree(it->second.p);
    g_splitk.erase(it);
}

// (256 x 256: 1 workgroup per CU; 128 x 256: 1 per CU, 0.68 of the time; 128 x 128: 2 per CU, 0.75 -- 0.41 alone)
template <int DT, int EPI>
static int launch_gemm_big(const GmArgs &P0, hipStream_t s)
{
    constexpr int CAP = (DT == F16) ? 32 : block_capacity(DT);
    GmArgs P = P0;
    const size_t T = (size_t)P.T, cus = (size_t)gemm_num_cus();
    auto ntiles = [&](size_t bn) {
        const size_t bne = EPI == GM_GLU ? bn / 2 : bn;
        size_t n = 0;
        for (int i = 0; i < P.nsets; i++) n += ifa_cdiv((size_t)P.rows[i], bne);
        return n;
    };
    auto run = [&](auto bm, auto bn, auto wm, auto wn, int tn0, int tn_count, auto bk, auto ks) -> int {      // weight tiles [tn0, tn0 + tn_count)
        constexpr int BM = decltype(bm)::value, BN = decltype(bn)::value, WM = decltype(wm)::value, WN = decltype(wn)::value, BK = decltype(bk)::value, KS = decltype(ks)::value;
        constexpr int BNE = EPI == GM_GLU ? BN / 2 : BN;
        BigGeo G;
        G.tile0[0] = 0;
        for (int i = 0; i < 3; i++) G.tile0[i + 1] = G.tile0[i] + (i < P.nsets ? (int)ifa_cdiv((size_t)P.rows[i], (size_t)BNE) : 0);
        for (int i = P.nsets; i < 3; i++) G.tile0[i] = 1 << 30;        // (absent sets are never selected)
        G.tiles_m = (int)ifa_cdiv(T, (size_t)BM); G.K = P.nblk * CAP; G.tn0 = tn0; G.part = nullptr; G.flags = nullptr; G.err = nullptr;
        G.sk_full = G.sk_rem = G.sk_tiles_n = G.sk_slots = 0;
        G.dbg_skip = (g_gemm_big >> 14) & 1;
        size_t grid = (size_t)G.tiles_m * tn_count * (KS ? KS : 1);
        if constexpr (KS == 0) {
            // stream-K: one workgroup per CU; whole tiles first, then equal shares of the K steps of the tiles that are left
            const size_t tiles = (size_t)G.tiles_m * tn_count, S = (size_t)G.K / BK;
            grid = std::min(cus, tiles * S);
            G.sk_full = (int)(tiles / grid); G.sk_rem = (int)(tiles % grid); G.sk_tiles_n = tn_count;
            const size_t lmin = (size_t)G.sk_rem * S / grid;
            G.sk_slots = G.sk_rem && lmin ? (int)ifa_cdiv(S - 1, lmin) : 0;
            if (G.sk_rem) {
                G.err = wait_err_word();
                void *scratch = nullptr;
                int rcs = gemm_splitk_scratch(s, (size_t)G.sk_rem * G.sk_slots * BM * BN * 4, (size_t)G.sk_rem, &scratch);
                if (rcs) return rcs;
                G.flags = (unsigned *)scratch; G.part = (unsigned long long *)((char *)scratch + SPLITK_FLAG_BYTES);
            }
        }
        if constexpr (KS > 1) {
            G.err = wait_err_word();
            const size_t tiles = (size_t)G.tiles_m * tn_count, part_bytes = tiles * (size_t)(KS - 1) * BM * BN * 4;
            void *scratch = nullptr;
            int rcs = gemm_splitk_scratch(s, part_bytes, tiles, &scratch);
            if (rcs) return rcs;
            G.flags = (unsigned *)scratch; G.part = (unsigned long long *)((char *)scratch + SPLITK_FLAG_BYTES);      // (flags at a FIXED place: they are zero between launches)
        }
        const size_t smem = std::max(2 * (size_t)(BM + BN) * (BK * 2), (size_t)BM * (BN * 2 + 64));      // operand tiles; the epilogue's output tile
        auto kern = k_gemm_big<DT, BM, BN, WM, WN, EPI, BK, KS>;
        // (function attributes are per device: one bit per device of this instantiation -- the C++ engine drives several GPUs
        // from one process)
        static std::atomic<uint64_t> attr_set{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            attr_set.fetch_or(bit, std::memory_order_relaxed);
        }
        if constexpr (KS != 1) {      // the second halves wait for the first halves: the whole grid must be resident at once (1: not launched)
            if (!wait_grid_fits((const void *)kern, WM * WN * 64, smem, (long long)grid)) return 1;
        }
        kern<<<dim3((unsigned)grid), dim3(WM * WN * 64), smem, s>>>(P, G);
        return IFA_OK;
    };
    using std::integral_constant;
    typedef integral_constant<int, 256> I256; typedef integral_constant<int, 128> I128; typedef integral_constant<int, 2> I2; typedef integral_constant<int, 4> I4; typedef integral_constant<int, 64> K64;
    typedef integral_constant<int, 1> S1;
    const int force = (g_gemm_big >> 8) & 7;        // (measurement: 1 / 2 / 3 force a tile shape; 4 / 5 / 6: the small tiles below)
    auto rounds = [&](size_t bm, size_t bn, size_t per_cu) { return (double)ifa_cdiv(ifa_cdiv(T, bm) * ntiles(bn), cus * per_cu); };
    // 256 x 256: whole rounds of the chip; what is left of the last round goes to a second launch of 128-token tiles when
    // those fit the chip in one go (1024 tokens x 11008 GLU pairs: 86 weight tiles = 64 x 4 workgroups + 22 x 8)
    const size_t tm256 = ifa_cdiv(T, (size_t)256), tn256 = ntiles(256), per_round = tm256 <= cus ? cus / tm256 : 0;
    const size_t full_n = per_round ? (tn256 / per_round) * per_round : 0, rem_n = tn256 - full_n;
    const bool split = full_n > 0 && rem_n > 0 && ifa_cdiv(T, (size_t)128) * rem_n <= cus && tm256 * rem_n < cus * 3 / 4;
    const size_t n128 = ifa_cdiv(T, (size_t)128) * ntiles(128);
    double c256 = CAP > 32 ? 1e30 : (split ? (double)(full_n / per_round) + 0.68 : rounds(256, 256, 1));   // (64-value blocks: the 256 x 256 variant would spill)
    // Round 6, stream-K on the 256 x 256 tiles (k_gemm_big<.., KS = 0>; OPT-IN, bit 13 of ifa_gemm_big_tiles): every CU takes
    // floor(tiles / CUs) whole tiles and then an equal share of the K steps of the rest.  Built, deterministic, within the F16 rounding of
    // the whole-tile kernel -- and measured SLOWER on the two products it was built for (1024-token prompt, Llama-2-7B Q4, one box,
    // rocprofv3 averages over 128 launches): w1 | w3 (344 tiles) 259.5 us against 144.3 + 84.2 for the round of whole tiles + the
    // 128-token-tile launch of the rest, wq | wk | wv (192 tiles) 135.6 against 117.5; prefill 20.4 against 18.6 ms (2048 tokens: 37.7
    // against 34.9).  The tail it removes is 35 / 29 us per layer; a share that does not end its tile sends 256 KB of fp32 sums through
    // memory (write-through stores, 64 MB per launch) and the share that ends the tile reads up to three of them -- MI355X_MICROARCH.md
    // prices that at ~12 us out + 4 us per slot in on an idle chip, here it costs 50-65.  Eligible when a share is >= 20 steps (<= 3
    // scratch slots per shared tile at K = 4096).  profiles/r06_stream_k_ab.log.
    bool sk = false;
    {
        const size_t tiles = tm256 * tn256, S = (size_t)P.nblk * CAP / PF_BK, rem = tiles % cus, share = rem * S / cus;
        const bool may_sk = !force && CAP <= 32 && (g_gemm_big & (1 << 13)) != 0 && !P.no_waits && waits_enabled() && tiles * 4 >= cus * 3;
        if (may_sk && rem && share >= 20 && ifa_cdiv(S - 1, share) <= 3) {
            const double csk = (double)(tiles / cus) + (double)rem / (double)cus + 0.10;
            if (csk < c256) { c256 = csk; sk = true; }
        }
    }
    const double c128x256 = rounds(128, 256, 1) * 0.68;
    const double c128 = n128 <= cus ? 0.41 : rounds(128, 128, 2) * 0.75;
    int pick = force;
    if (!pick) pick = (c256 <= c128x256 && c256 <= c128) ? 1 : (c128x256 <= c128 ? 2 : 3);
    int rc = IFA_OK;
    typedef integral_constant<int, 64> I64;
    if (pick == 4 && CAP <= 32) rc = run(I64(), I64(), I2(), I2(), 0, (int)ntiles(64), K64(), S1());
    else if (pick == 5 && CAP <= 32) rc = run(I128(), I64(), I2(), I2(), 0, (int)ntiles(64), K64(), S1());
    else if (pick == 6 && CAP <= 32) rc = run(I64(), I128(), I2(), I2(), 0, (int)ntiles(128), K64(), S1());
    else if (pick == 1 && CAP <= 32) {
        rc = 1;
        if (sk) rc = run(I256(), I256(), I2(), I4(), 0, (int)tn256, K64(), integral_constant<int, 0>());
        if (rc != 1) { /* stream-K launched (or failed for another reason than residency) */ }
        else if (split && !force) {
            rc = run(I256(), I256(), I2(), I4(), 0, (int)full_n, K64(), S1());
            if (!rc) rc = run(I128(), I256(), I2(), I4(), (int)full_n, (int)rem_n, K64(), S1());
        } else rc = run(I256(), I256(), I2(), I4(), 0, (int)tn256, K64(), S1());
    } else if (pick == 2 || pick == 1)
        rc = run(I128(), I256(), I2(), I4(), 0, (int)tn256, K64(), S1());
    else {
        // 128 x 128 tiles that fill no more than one workgroup slot per CU (two fit): both halves of K at once -- only for products
        // big enough to matter (>= half the chip), never when a tile shape is forced (measurement / the bit-identity tests).
        // Llama-2-7B prefill: 1024 tokens 51.4K -> 54.2K tok/s, 768: 40.4K -> 44.2K, 512: 35.1K -> 42.4K.  (256 x 256 tiles split four
        // ways -- the efficient tile shape, one part per CU -- measured no better: 53.2K at 1024 tokens, three partial tiles through memory.)
        const bool may_split = !force && EPI != GM_GLU && (g_gemm_big & (1 << 12)) == 0 && !P.no_waits && waits_enabled();
        const size_t ksteps = (size_t)P.nblk * CAP / PF_BK;
        const bool splitk = may_split && n128 <= cus && n128 * 2 >= cus && ksteps % 2 == 0 && P.nblk * CAP >= 2048;
        // Round 5, prompts of 48..128 tokens (ONE token tile; the engine sends them here from `prefill_big_min` + 1 tokens on): wq | wk | wv
        // are 96 tiles, wo and w2 32 -- a quarter or an eighth of the chip.  Four parts of K per tile when that still fits two
        // workgroups per CU and a part keeps >= 1024 columns: 64 tokens 7.46 -> 7.10 ms, 128 tokens 8.22 -> 7.46 (profiles/r05_prompt_lengths.log)
        const bool splitk4 = may_split && n128 * 2 < cus && n128 * 4 <= 2 * cus && ksteps % 4 == 0 && P.nblk * CAP >= 4096;
        // Eight waves per tile (2 x 4: two per SIMD take turns in the step's serial chain) up to 512 tokens: 64 / 128 / 256 / 512 tokens
        // 7.17 / 7.44 / 8.60 / 12.28 -> 6.86 / 7.09 / 8.47 / 12.15 ms, three alternating runs on one box; four waves above (1024 tokens:
        // 19.0 vs 19.25 ms -- there the tiles are wo / w2 in two halves of K, two workgroups per CU already).  Same products either way.
        const bool w8 = !force && T <= 512 && CAP <= 32;
        auto run128 = [&](auto ks) -> int {
            if (w8) return run(I128(), I128(), I2(), I4(), 0, (int)ntiles(128), K64(), ks);
            return run(I128(), I128(), I2(), I2(), 0, (int)ntiles(128), K64(), ks);
        };
        if constexpr (EPI != GM_GLU) {
            rc = 1;
            if (splitk4) rc = run128(integral_constant<int, 4>());
            else if (splitk) rc = run128(integral_constant<int, 2>());
            if (rc == 1) rc = run128(S1());      // (no split, or its grid cannot be resident at once)
        } else rc = run128(S1());      // (K steps of 128 columns -- BK = 128 -- measured: 57 -> 72 us at 1024 x 4096 x 4096)
    }
    if (rc) return rc;
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}


template <int DT>
static int launch_gemm(const void *W, size_t N, size_t K, const void *X, size_t T, const void *bias, void *Y, hipStream_t s)
{
    constexpr int CAP = (DT == F16) ? 32 : block_capacity(DT);
    const int nblk = (int)(K / CAP);
    constexpr int MT = 2;
    const size_t slab = (size_t)32 * MT * (2 * CAP * 2 + 16);
    if (T > 128 && K % PF_BK == 0 && CAP <= 32 && PF_BK % CAP == 0 && N % 8 == 0 && (g_gemm_big & 1)) {
        GmArgs P; memset(&P, 0, sizeof(P));
        P.W[0] = (const uint8_t *)W; P.rows[0] = (int)N; P.nsets = 1; P.nblk = nblk; P.T = (int)T;
        P.X = (const half_t *)X; P.ldx = (int)K; P.bias[0] = (const half_t *)bias; P.Y = (half_t *)Y; P.ldy = (int)N;
        return launch_gemm_big<DT, GM_PLAIN>(P, s);
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

namespace ifa {
// cols % 64 == 0, 64 % block capacity == 0; fused epilogues (GM_RESIDUAL / GM_GLU) on the 20-byte Q4 blocks
bool gemm_big_ok(int w_dtype, const GmArgs &P, int epi)
{
    const int cap = w_dtype == F16 ? 32 : block_capacity(w_dtype);
    if (cap <= 1 || PF_BK % cap != 0 || ((size_t)P.nblk * cap) % PF_BK != 0 || P.T < 1 || P.nsets < 1 || P.nsets > 3) return false;
    if (epi != GM_PLAIN && w_dtype != Q4_B32T1A && w_dtype != Q4_B32T1B) return false;
    if (epi == GM_GLU && (P.nsets != 1 || !P.W1)) return false;
    for (int i = 0; i < P.nsets; i++) if (P.rows[i] % 8 != 0) return false;     // (16-byte pieces of output rows)
    if ((P.Yset[0] ? (P.ldyset[0] | P.ldyset[1] | P.ldyset[2]) : P.ldy) % 8 != 0 || (epi == GM_RESIDUAL && P.ldres % 8 != 0)) return false;
    return true;
}
int gemm_big(int w_dtype, const GmArgs &P, int epi, hipStream_t s)
{
    if (!gemm_big_ok(w_dtype, P, epi)) return ifa_fail(IFA_ERR_ARG, "large-tile GEMM: dtype %d / %d blocks per row / epilogue %d", w_dtype, P.nblk, epi);
    if (epi == GM_PLAIN) {
        if (w_dtype == F16) return launch_gemm_big<F16, GM_PLAIN>(P, s);
        IFA_DISPATCH_QUANT_DTYPE(w_dtype, return (launch_gemm_big<DT, GM_PLAIN>(P, s)));
        return IFA_OK;
    }
    if (w_dtype == Q4_B32T1A) return epi == GM_RESIDUAL ? launch_gemm_big<Q4_B32T1A, GM_RESIDUAL>(P, s) : launch_gemm_big<Q4_B32T1A, GM_GLU>(P, s);
    return epi == GM_RESIDUAL ? launch_gemm_big<Q4_B32T1B, GM_RESIDUAL>(P, s) : launch_gemm_big<Q4_B32T1B, GM_GLU>(P, s);
}
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
    // (every T runs the in-tree kernels: no vendor GEMM is linked or loaded by this library; the dequantise-once + hipBLASLt
    //  comparison of rounds 1-3 is tools/bench_gemm_big.py, through torch.matmul on the dequantised operand)
    if (w_dtype == F16) {
        launch_gemm<F16>(W, rows, cols, x_f16, tokens, bias_f16, y_f16, s);
    } else {
        IFA_DISPATCH_QUANT_DTYPE(w_dtype, launch_gemm<DT>(W, rows, cols, x_f16, tokens, bias_f16, y_f16, s));
    }
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// frees the per-stream scratch the prefill kernels keep (the attention score-tile workspace) on the current device
extern "C" int ifa_gemm_release_stream(ifa_stream stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return IFA_OK;
    (void)hipStreamSynchronize(ifa_s(stream));
    ifa::attn_release_stream(dev, ifa_s(stream));
    gemm_splitk_release(dev, ifa_s(stream));
    ifa::rows_kparts_release(dev, ifa_s(stream));
    return IFA_OK;
}
