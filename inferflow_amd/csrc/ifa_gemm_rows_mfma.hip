// ifa_gemm_rows_mfma.hip -- Y[T][N] = X[T][K] . W[N][K]^T for 2 <= T <= 16 rows (dynamic batching of decode steps, very
// short prompts, MoE experts with a handful of rows) on v_mfma_f32_16x16x32_f16, streaming the tiled Q4_B32T1 weights
// ONCE.  With the prologue / epilogues of the fused batched decode step (GmArgs: RMS norm while staging, wq | wk | wv as one
// virtual row space, w1 / w3 as interleaved tile pairs with act(.) * (.), residual add) and a grouped variant for MoE experts.
//
// Same contract as ifa_gemm / ifa_gemm_rows_q4 (the reference's T > 1 branch: weights dequantised to half, half
// activations, fp32 accumulation, one F16 rounding, bias as a half add: MatrixMultiplication,
// src/transformer/inference_worker.cc:2374-2415).  The fdot2 kernel of ifa_gemm_rows.hip spends T/2 + 3 VALU operations
// per weight and is VALU-bound at ~1.7-2 TB/s; here a weight costs its dequantisation only (the products run on the
// matrix cores), and a 16 x 16 tile wastes little of them at 2..16 rows.
//
// Work decomposition (one workgroup of 8 waves per CU):
//   * tile = 16 consecutive weight rows; the K range of a tile is SPLIT over the 8 waves -- within a chunk of the activation
//     rows (4096 columns, or 2048 for 9..16 rows: GmGeo) wave w takes blocks BPW w .. BPW w + BPW - 1 (BPW = 16 or 8) -- so
//     every wave of the chip has requests in flight from the first instruction (a 4096-row matrix is only 256 tiles);
//     partial 16 x 16 tiles are summed through LDS in wave order (deterministic);
//   * a wave's share of a tile and chunk is 16 rows x BPW blocks.  It is REQUESTED coalesced -- BPW * 16 contiguous code
//     bytes per row, 64 / BPW rows per request (+ one request for the (base, scale) words) -- because 16-row x 64-byte
//     requests (each lane its own MFMA operand) ran at 2.4 TB/s against 3.5 TB/s for contiguous ones; the wave then
//     turns the group through its own LDS patch (no barrier: one wave) into the MFMA layout: lane (r = lane % 16,
//     g = lane / 16) reads block 4s + g of row r.  The block's four 8-element quarters feed FOUR MFMAs as k-group g:
//     MFMA q of a superstep multiplies the columns {32 (4s + g) + 8 q .. + 8 : g = 0..3} -- the B fragments are read from
//     LDS with the same permutation, so the sum over k is the plain dot product in another (fixed) order;
//   * the activation rows sit in LDS as F16, one chunk at a time (row stride + 16 B: conflict-free 16-byte reads); longer
//     rows (w2) are walked in chunks with the accumulators kept in registers.
// Measurements behind these choices and what bounds the kernel now: DESIGN.md section 3 "Dynamic batching", section 8.
#include <algorithm>
#include <cstring>
#include "ifa_host.h"
#include "ifa_decode_kernels.h"
#include "ifa_moe.h"
#include "ifa_gemm_rows_mfma.h"

namespace ifa {

typedef _Float16 h8m __attribute__((ext_vector_type(8)));
typedef float f4m __attribute__((ext_vector_type(4)));
typedef float f2m __attribute__((ext_vector_type(2)));
typedef _Float16 h2m __attribute__((ext_vector_type(2)));

constexpr int GM_THREADS = 512, GM_WAVES = 8;
// Geometry by CS = supersteps (128 columns) per LDS chunk of the activation rows: 32 (4096 columns: up to 8 rows fit next to
// the waves' patches) or 16 (2048 columns: up to 16 rows).  A wave's share of a chunk is BPW = CS / 2 blocks per row.
template <int CS> struct GmGeo {
    static constexpr int CHUNK_SUP = CS;
    static constexpr int CHUNK_COLS = CS * 128;
    static constexpr int ROW_STRIDE = CHUNK_COLS * 2 + 16;      // bytes per activation row in LDS (+16: conflict-free 16-byte reads)
    static constexpr int BPW = CS * 4 / GM_WAVES;               // blocks of a chunk per wave and row: 16 or 8
    static constexpr int NJ = BPW / 4;                          // supersteps per group
    static constexpr int NI = BPW / 4;                          // code requests per group: 64 lanes cover 64 / BPW rows x BPW blocks
    static constexpr int CSTRIDE = BPW * 16 + 16;               // patch: bytes per row of code blocks
    static constexpr int SSTRIDE = BPW * 4 + 16;                // patch: bytes per row of (base, scale) words
    static constexpr int PATCH_BYTES = 16 * CSTRIDE + 16 * SSTRIDE;
    static constexpr int PIECES = CHUNK_COLS / 8;               // 16-byte pieces per staged row: 512 or 256
    static constexpr int XR = GM_THREADS / PIECES;              // rows staged side by side: 1 or 2
};
constexpr int gm_cs(int tx) { return tx > 8 ? 16 : 32; }
template <int NI> struct GmGrpT { u32x4 c[NI]; u32x4 sb; };

struct GmTile { const uint8_t *W0; const half_t *b0; half_t *y; int row0, nrows, vrow0, ldy; };
// The set of a tile is selected among SCALARS read once from the argument block (GmSets): selecting among the struct's
// fields in place made the compiler spill the whole block to scratch and fetch the chosen field with a VGPR-indexed
// scratch load in front of every weight request (first version of the fused step: every kernel +6 us).
struct GmSets { const uint8_t *w0, *w1, *w2; const half_t *b0, *b1, *b2; half_t *y0, *y1, *y2; int r0, r1, r2, nsets, l0, l1, l2; };
// (`c ? S.a : S.b` on two members is an lvalue conditional: clang selects the ADDRESS and loads once -- through scratch with
//  a VGPR index when the struct is a local.  gm_sel takes its operands by value, so the select is on values.)
template <typename V> __device__ __forceinline__ V gm_sel(bool c, V a, V b) { return c ? a : b; }
__device__ __forceinline__ GmTile gm_locate(const GmSets &S, int vt)
{
    const uint8_t *const w0 = S.w0, *const w1 = S.w1, *const w2 = S.w2;
    const half_t *const b0 = S.b0, *const b1 = S.b1, *const b2 = S.b2;
    const int r0 = S.r0, r1 = S.r1, r2 = S.r2;
    const int t0 = (r0 + 15) >> 4, t1 = (r1 + 15) >> 4;
    const bool in1 = S.nsets > 1 && vt >= t0, in2 = S.nsets > 2 && vt >= t0 + t1;
    GmTile t;
    t.W0 = gm_sel(in2, w2, gm_sel(in1, w1, w0));
    t.b0 = gm_sel(in2, b2, gm_sel(in1, b1, b0));
    t.nrows = gm_sel(in2, r2, gm_sel(in1, r1, r0));
    half_t *const y0 = S.y0, *const y1 = S.y1, *const y2 = S.y2;
    const int l0 = S.l0, l1 = S.l1, l2 = S.l2;
    t.y = gm_sel(in2, y2, gm_sel(in1, y1, y0));           // the set's output matrix (row 0 of the set) and its row stride
    t.ldy = gm_sel(in2, l2, gm_sel(in1, l1, l0));
    const int lt = gm_sel(in2, vt - t0 - t1, gm_sel(in1, vt - t0, vt));
    t.row0 = lt * 16;
    t.vrow0 = gm_sel(in2, r0 + r1, gm_sel(in1, r0, 0)) + lt * 16;
    return t;
}

// MAXT: tiles per workgroup (tile = blockIdx.x + i * gridDim.x); TX: activation rows staged per thread (>= T, power of two)
// EPI: GmEpilogue; NORM: 1 = RMS-normalise the rows while staging them (K <= 4096: one chunk)
// Every global load below is UNCONDITIONAL (clamped or redirected addresses): loads inside branches make the compiler's
// vmcnt bookkeeping conservative -- every wait became vmcnt(0), i.e. for all groups in flight (ISA of the first version).
template <int MAXT, int TX, int EPI, int NORM>
__device__ __forceinline__ void gemm_rows_mfma_body(const GmArgs &P, char *smem)
{
    // GM_GLU: a workgroup's tiles come in PAIRS -- `it` even: tile (it / 2) of w1, odd: the same tile of w3 -- so the gated
    // product keeps the plain kernel's registers and prefetch depth (MAXT counts both; the epilogue pairs the accumulators)
    constexpr bool GLU = EPI == GM_GLU;
    static_assert(!GLU || MAXT % 2 == 0, "GM_GLU: tiles per workgroup come in pairs");
    using G = GmGeo<gm_cs(TX)>;
    using GmGrp = GmGrpT<G::NI>;
    static_assert(NORM == 0 || G::CHUNK_SUP == 32, "the norm prologue needs the whole row in one chunk");
    constexpr int PD = 3;                             // groups in flight per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int nblk = P.nblk, T = P.T;
    // (readfirstlane: the values are materialised in SGPRs here, so the selects below cannot be folded back into a load
    //  through a selected ADDRESS of the argument block)
    auto sp = [](const void *p) {
        const uint64_t v = (uint64_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    auto si = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    // outputs: one matrix over the virtual rows (Y, ldy), or one matrix per set (Yset, ldyset)
    const bool per_set = P.Yset[0] != nullptr;
    half_t *const ya = per_set ? P.Yset[0] : P.Y, *const yb = per_set ? P.Yset[1] : P.Y + P.rows[0],
           *const yc = per_set ? P.Yset[2] : P.Y + P.rows[0] + P.rows[1];
    const GmSets S = {(const uint8_t *)sp(P.W[0]), (const uint8_t *)sp(P.W[1]), (const uint8_t *)sp(P.W[2]), (const half_t *)sp(P.bias[0]),
                      (const half_t *)sp(P.bias[1]), (const half_t *)sp(P.bias[2]), (half_t *)sp(ya), (half_t *)sp(yb), (half_t *)sp(yc),
                      si(P.rows[0]), si(P.rows[1]), si(P.rows[2]), si(P.nsets),
                      si(per_set ? P.ldyset[0] : P.ldy), si(per_set ? P.ldyset[1] : P.ldy), si(per_set ? P.ldyset[2] : P.ldy)};
    const uint8_t *const W1p = P.W1;
    const half_t *const Xp = P.X, *const nwp = P.norm_w;
    const int ldx = P.ldx;
    const int K = nblk * 32;
    const int nsup = nblk >> 2;                                   // nblk % 4 == 0 (checked by the launcher)
    const int nchunk = (nsup + G::CHUNK_SUP - 1) / G::CHUNK_SUP;
    const int ntiles = (P.total_rows + 15) >> 4;                  // (several sets: every set is whole tiles)
    const size_t row_bytes = tiled_row_bytes(Q4_B32T1A, (size_t)nblk);
    const int trow = min(r, T - 1);                               // B operand: token of this lane (columns past T: duplicates, never stored)

    // a group = this wave's 16 rows x BPW blocks of (tile, chunk): blocks blk0 .. blk0 + BPW - 1, blk0 = 4 CS chunk + BPW wave
    // valid == false (past the last group): all lanes re-read the first bytes of the matrix -- one cache line, no branch
    auto tile_of = [&](int it) { return GLU ? (int)blockIdx.x + (it >> 1) * (int)gridDim.x : (int)blockIdx.x + it * (int)gridDim.x; };
    constexpr int RPI = 64 / G::BPW;               // rows per code request
    constexpr int LPR = G::BPW / 4;                // lanes per row of (base, scale) words (4 blocks' words each)
    auto fetch = [&](GmGrp &q, int it, int chunk, bool valid) {
        const GmTile tl = gm_locate(S, min(tile_of(it), ntiles - 1));
        const int blk0 = chunk * (G::CHUNK_SUP * 4) + wave * G::BPW;
        const uint8_t *Wt = gm_sel(GLU && (it & 1), W1p, tl.W0);       // (by value: see gm_sel)
#pragma unroll
        for (int i = 0; i < G::NI; i++) {      // codes: lane l -> row RPI i + l / BPW, block l % BPW (BPW * 16 contiguous bytes per row)
            const int row = valid ? min(tl.row0 + RPI * i + lane / G::BPW, tl.nrows - 1) : 0;
            const int blk = valid ? min(blk0 + (lane % G::BPW), nblk - 1) : 0;
            q.c[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(Wt + (size_t)row * row_bytes + (size_t)blk * 16));
        }
        {                                       // (base, scale): 16 rows x LPR lanes, four blocks' words per lane (upper lanes: duplicates)
            const int ls = lane & (16 * LPR - 1);
            const int row = valid ? min(tl.row0 + ls / LPR, tl.nrows - 1) : 0;
            const int blk = valid ? min(blk0 + 4 * (ls % LPR), nblk - 4) : 0;
            q.sb = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(Wt + (size_t)row * row_bytes + (size_t)nblk * 16 + (size_t)blk * 4));
        }
    };
    char *patch = smem + (size_t)TX * G::ROW_STRIDE + (size_t)wave * G::PATCH_BYTES;        // this wave's transposition patch
    auto compute = [&](const GmGrp &q, int chunk, f4m &acc) {
        // ---- through the patch: rows of BPW code blocks and rows of BPW (base, scale) words, both at a stride that makes the
        // 16-row reads below conflict-free
#pragma unroll
        for (int i = 0; i < G::NI; i++)
            *reinterpret_cast<u32x4 *>(patch + (size_t)(RPI * i + lane / G::BPW) * G::CSTRIDE + (size_t)(lane % G::BPW) * 16) = q.c[i];
        if (lane < 16 * LPR) *reinterpret_cast<u32x4 *>(patch + 16 * G::CSTRIDE + (size_t)(lane / LPR) * G::SSTRIDE + (size_t)(lane % LPR) * 16) = q.sb;
        const int blk0 = chunk * (G::CHUNK_SUP * 4) + wave * G::BPW;
#pragma unroll
        for (int j = 0; j < G::NJ; j++) {
            if (blk0 + 4 * j >= nblk) continue;                    // wave-uniform: past the row end (nblk % 4 == 0)
            const u32x4 cw4 = *reinterpret_cast<const u32x4 *>(patch + (size_t)r * G::CSTRIDE + (size_t)(4 * j + g) * 16);
            const uint32_t sbw = *reinterpret_cast<const uint32_t *>(patch + 16 * G::CSTRIDE + (size_t)r * G::SSTRIDE + (size_t)(4 * j + g) * 4);
            const float base = hbits2f((uint16_t)(sbw & 0xFFFFu)), scale = hbits2f((uint16_t)(sbw >> 16));
            const char *xrow = smem + (size_t)trow * G::ROW_STRIDE + (size_t)((wave * G::BPW + 4 * j + g) * 32) * 2;
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const uint32_t cw = cw4[s4];
                // byte b of the word: low nibble = element 8 s4 + 2b, high nibble = the next one (ifa_gemm_rows.hip)
                const uint32_t lo = cw & 0x0F0F0F0Fu, hi = (cw >> 4) & 0x0F0F0F0Fu;
                // two weights per v_pk_fma_f32 + v_cvt_pk_f16_f32 (round to nearest even): the reference's dequantised halves
                const f2m s2 = {scale, scale}, b2 = {base, base};
                const f2m q0 = {ubyte_f32<0>(lo), ubyte_f32<0>(hi)}, q1 = {ubyte_f32<1>(lo), ubyte_f32<1>(hi)};
                const f2m q2 = {ubyte_f32<2>(lo), ubyte_f32<2>(hi)}, q3 = {ubyte_f32<3>(lo), ubyte_f32<3>(hi)};
                const h2m w0 = __builtin_convertvector(__builtin_elementwise_fma(q0, s2, b2), h2m);
                const h2m w1 = __builtin_convertvector(__builtin_elementwise_fma(q1, s2, b2), h2m);
                const h2m w2 = __builtin_convertvector(__builtin_elementwise_fma(q2, s2, b2), h2m);
                const h2m w3 = __builtin_convertvector(__builtin_elementwise_fma(q3, s2, b2), h2m);
                const h8m a = {w0[0], w0[1], w1[0], w1[1], w2[0], w2[1], w3[0], w3[1]};
                const h8m b = *reinterpret_cast<const h8m *>(xrow + s4 * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            }
        }
    };

    f4m acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; i++) acc[i] = f4m{0.0f, 0.0f, 0.0f, 0.0f};
    // PD groups of requests in flight per wave (a group = 4 supersteps = 5 KB per wave): one group ahead left every wave
    // waiting ~half of the time for HBM
    GmGrp buf[PD];
    const int nq = nchunk * MAXT;                                  // (chunk, tile) pairs in execution order: chunk outer
    auto fetch_q = [&](GmGrp &q, int qi) { const int ch = qi / MAXT; fetch(q, qi - ch * MAXT, ch, qi < nq); };
    // The CU's memory pipeline is FIFO across waves (ifa_decode_kernels.h): the activation rows of the first chunk are
    // requested by all threads, and a barrier passed, BEFORE any weight request -- else they arrive behind the weights.
    // A full chunk is 512 16-byte pieces per row: piece tid of row k is thread tid's k-th request.
    // XR rows are staged side by side: thread tid holds piece tid % PIECES of rows tid / PIECES + XR k
    constexpr int XK = (TX + G::XR - 1) / G::XR;
    const int xpiece = tid % G::PIECES, xsub = tid / G::PIECES;
    u32x4 xv[XK];
    u32x4 nwv = {0, 0, 0, 0};
    auto x_request = [&](int chunk) {
        const int c0 = chunk * G::CHUNK_COLS;
        const int per_row = min(G::CHUNK_COLS, K - c0) >> 3;
#pragma unroll
        for (int k = 0; k < XK; k++)       // rows past T and pieces past the row end: clamped (duplicates), never stored
            xv[k] = *reinterpret_cast<const u32x4 *>(Xp + (size_t)min(xsub + G::XR * k, T - 1) * ldx + c0 + (size_t)min(xpiece, per_row - 1) * 8);
        if constexpr (NORM == 1)
            nwv = *reinterpret_cast<const u32x4 *>((nwp ? nwp : Xp) + (size_t)min(xpiece, per_row - 1) * 8);   // (no weight: a valid dummy address)
    };
    auto x_store = [&](int chunk) {
        const int per_row = min(G::CHUNK_COLS, K - chunk * G::CHUNK_COLS) >> 3;
        if constexpr (NORM == 1) {
            // RMS norm of every row in the canonical order of ifa_math.h: piece c = tid is lane c % 64 of group c / 64 = wave
            // (NORM variants stage one row per pass: XR == 1)
            float *part = reinterpret_cast<float *>(smem + (size_t)TX * G::ROW_STRIDE + (size_t)GM_WAVES * G::PATCH_BYTES);     // [TX][8]
#pragma unroll
            for (int k = 0; k < XK; k++) {
                rms_h8 v8 = __builtin_bit_cast(rms_h8, xv[k]);
                if (tid >= per_row) {
#pragma unroll
                    for (int i = 0; i < 8; i++) v8[i] = (half_t)0;
                }
                const float pg = wave_sum(rms_chunk_sq(v8));
                if (lane == 0) part[k * GM_WAVES + wave] = pg;
            }
            __syncthreads();
            const rms_h8 nw8 = __builtin_bit_cast(rms_h8, nwv);
#pragma unroll
            for (int k = 0; k < XK; k++) {
                const float scale = rms_scale_of(rms_total(part + k * GM_WAVES, (per_row + 63) >> 6), K, P.eps);
                const rms_h8 v8 = __builtin_bit_cast(rms_h8, xv[k]);
                rms_h8 o;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float t = (float)v8[i] * scale;
                    if (nwp) { const float mlt = P.multi_base + (float)nw8[i]; t = t * mlt; }
                    o[i] = f2h(t);
                }
                xv[k] = __builtin_bit_cast(u32x4, o);
            }
        }
#pragma unroll
        for (int k = 0; k < XK; k++) {
            const int row = xsub + G::XR * k;
            if (row < T && xpiece < per_row) *reinterpret_cast<u32x4 *>(smem + (size_t)row * G::ROW_STRIDE + (size_t)xpiece * 16) = xv[k];
        }
    };
    x_request(0);
    __syncthreads();
#pragma unroll
    for (int d = 0; d < PD; d++) fetch_q(buf[d], d);
    for (int chunk = 0; chunk < nchunk; chunk++) {
        if (chunk > 0) {
            __syncthreads();                                       // the previous chunk's fragments have been read
            x_request(chunk);
        }
        x_store(chunk);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < MAXT; it++) {
            const int qi = chunk * MAXT + it;
            compute(buf[0], chunk, acc[it]);
#pragma unroll
            for (int d = 0; d + 1 < PD; d++) buf[d] = buf[d + 1];
            fetch_q(buf[PD - 1], qi + PD);
        }
    }
    // ---- sum the 8 waves' partial tiles in wave order, then the epilogue: thread e of the first 256 owns element
    // (m = (l >> 4) * 4 + i, n = l & 15) of every tile, l = e >> 2, i = e & 3 (the MFMA's C layout)
    __syncthreads();                                               // the activation image is free: partials take its place
    float *part = reinterpret_cast<float *>(smem);                 // [MAXT][8 waves][256]
#pragma unroll
    for (int it = 0; it < MAXT; it++)
        *reinterpret_cast<f4m *>(part + ((size_t)(it * GM_WAVES + wave) * 64 + lane) * 4) = acc[it];
    __syncthreads();
    if (tid < 256) {
        const int l = tid >> 2, i = tid & 3;
        const int m = (l >> 4) * 4 + i, n = l & 15;
        auto total = [&](int it) {
            float sum = 0.0f;
#pragma unroll
            for (int w = 0; w < GM_WAVES; w++) sum = sum + part[(size_t)(it * GM_WAVES + w) * 256 + tid];
            return sum;
        };
        constexpr int STEP = GLU ? 2 : 1;
#pragma unroll
        for (int it = 0; it < MAXT; it += STEP) {
            const int vt = tile_of(it);
            const GmTile tl = gm_locate(S, min(vt, ntiles - 1));
            const int row = tl.row0 + m;
            const float s0 = total(it);
            float s1 = 0.0f;
            if constexpr (GLU) s1 = total(it + 1);
            if (vt < ntiles && row < tl.nrows && n < T) {
                half_t y = f2h(s0);
                if (tl.b0) y = f2h(h2f(y) + h2f(tl.b0[row]));
                const size_t vrow = (size_t)tl.vrow0 + m;
                if constexpr (EPI == GM_RESIDUAL) {
                    y = f2h(h2f(P.res[(size_t)n * P.ldres + vrow]) + h2f(y));          // TensorOpr::Add (half add)
                } else if constexpr (GLU) {
                    half_t y3 = f2h(s1);
                    if (P.bias1) y3 = f2h(h2f(y3) + h2f(P.bias1[row]));
                    const half_t act = f2h(act_fn(h2f(y), P.act_kind));                // TensorOpr::Activation -> F16
                    y = f2h(h2f(act) * h2f(y3));                                       // TensorOpr::Mul
                }
                tl.y[(size_t)n * tl.ldy + row] = y;
            }
        }
    }
}

template <int MAXT, int TX, int EPI, int NORM>
__global__ void __launch_bounds__(GM_THREADS) k_gemm_rows_mfma(const GmArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_rows_mfma_body<MAXT, TX, EPI, NORM>(P, smem);
}

// Mixture of experts (ifa_moe.h "smalls"): blockIdx.y is one expert's group of 2..8 consecutive rows of the gathered
// activations; its tiled weights come from the pointer table.
template <int MAXT, int TX>
__global__ void __launch_bounds__(GM_THREADS) k_gemm_rows_mfma_grouped(const MoeSmallGroup grp, int rows, int nblk, const half_t *__restrict__ X,
                                                                       half_t *__restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.y >= grp.counts[3]) return;
    const MoeTile gq = grp.smalls[blockIdx.y];
    GmArgs P;
    P.W[0] = grp.wtab_tiled[4 * gq.expert + grp.which_tiled]; P.W[1] = nullptr; P.W[2] = nullptr; P.W1 = nullptr;
    P.rows[0] = rows; P.rows[1] = 0; P.rows[2] = 0; P.nsets = 1; P.total_rows = rows; P.nblk = nblk; P.T = gq.nrows;
    P.X = X + (size_t)gq.row0 * nblk * 32; P.ldx = nblk * 32; P.multi_base = 0.0f; P.eps = 0.0f; P.norm_w = nullptr;
    P.bias[0] = nullptr; P.bias[1] = nullptr; P.bias[2] = nullptr; P.bias1 = nullptr;
    P.Y = Y + (size_t)gq.row0 * rows; P.res = nullptr; P.ldy = rows; P.ldres = 0; P.act_kind = 0;
    P.Yset[0] = nullptr; P.Yset[1] = nullptr; P.Yset[2] = nullptr; P.ldyset[0] = 0; P.ldyset[1] = 0; P.ldyset[2] = 0;
    gemm_rows_mfma_body<MAXT, TX, GM_PLAIN, 0>(P, smem);
}

static int gm_num_cus()
{
    static int n = 0;
    if (!n) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

static int gm_tx(int T) { return T <= 2 ? 2 : (T <= 4 ? 4 : (T <= 8 ? 8 : 16)); }
static size_t gm_smem(int T, int maxt)
{
    const int tx = gm_tx(T);
    const size_t row = tx > 8 ? GmGeo<16>::ROW_STRIDE : GmGeo<32>::ROW_STRIDE, patch = tx > 8 ? GmGeo<16>::PATCH_BYTES : GmGeo<32>::PATCH_BYTES;
    const size_t ximg = (size_t)tx * row + (size_t)GM_WAVES * patch + 8 * GM_WAVES * 4;      // + the norm's group sums
    const size_t parts = (size_t)maxt * GM_WAVES * 256 * 4;
    return std::max(ximg, parts);
}

bool gemm_rows_mfma_ok(size_t rows, size_t cols, size_t tokens)
{
    return tokens >= 2 && tokens <= 16 && cols % 128 == 0 && rows > 0 && rows < (1u << 24) && cols <= 65536;
}

static int gm_geometry(size_t rows, int grid_cap, int *wgs, int *maxt)
{
    const int ntiles = (int)((rows + 15) / 16);
    int w = std::min(grid_cap, ntiles);
    int mt = (ntiles + w - 1) / w;
    if (mt > 8) { mt = 8; w = (ntiles + 7) / 8; }                // (more workgroups than CUs: they queue)
    if (mt == 5) mt = 6;
    if (mt == 7) mt = 8;
    *wgs = w; *maxt = mt;
    return IFA_OK;
}

template <int MT, int TX, int EPI, int NORM>
static int gm_launch4(int wgs, size_t smem, const GmArgs &P, hipStream_t s)
{
    auto kern = k_gemm_rows_mfma<MT, TX, EPI, NORM>;
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)wgs), dim3(GM_THREADS), smem, s>>>(P);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

template <int MT, int TX>
static int gm_launch2(int wgs, size_t smem, const GmArgs &P, int epi, int norm, hipStream_t s)
{
    if (epi == GM_PLAIN && norm == 0) return gm_launch4<MT, TX, GM_PLAIN, 0>(wgs, smem, P, s);
    if (epi == GM_RESIDUAL && norm == 0) return gm_launch4<MT, TX, GM_RESIDUAL, 0>(wgs, smem, P, s);
    if constexpr (TX <= 8) {
        if (epi == GM_PLAIN && norm == 1) return gm_launch4<MT, TX, GM_PLAIN, 1>(wgs, smem, P, s);
        if constexpr (MT % 2 == 0) { if (epi == GM_GLU && norm == 1) return gm_launch4<MT, TX, GM_GLU, 1>(wgs, smem, P, s); }
    } else {
        if constexpr (MT % 2 == 0) { if (epi == GM_GLU && norm == 0) return gm_launch4<MT, TX, GM_GLU, 0>(wgs, smem, P, s); }
    }
    return ifa_fail(IFA_ERR_ARG, "rows GEMM: no kernel for epilogue %d / norm %d", epi, norm);
}

template <int MT>
static int gm_launch1(int wgs, size_t smem, const GmArgs &P, int epi, int norm, hipStream_t s)
{
    if (P.T <= 2) return gm_launch2<MT, 2>(wgs, smem, P, epi, norm, s);
    if (P.T <= 4) return gm_launch2<MT, 4>(wgs, smem, P, epi, norm, s);
    if (P.T <= 8) return gm_launch2<MT, 8>(wgs, smem, P, epi, norm, s);
    return gm_launch2<MT, 16>(wgs, smem, P, epi, norm, s);
}

bool gemm_rows_mfma_fused_ok(const GmArgs &P, int epi, int norm)
{
    if (P.T < 2 || P.T > 16 || P.nblk <= 0 || P.nblk % 4 != 0 || P.nsets < 1 || P.nsets > 3) return false;
    if (norm == 1 && (P.nblk * 32 > GmGeo<32>::CHUNK_COLS || P.T > 8)) return false;      // the norm prologue: whole rows in one chunk, <= 8 rows
    if (epi == GM_GLU && (P.nsets != 1 || !P.W1)) return false;
    for (int i = 0; i < P.nsets; i++) if (P.rows[i] <= 0 || (P.nsets > 1 && P.rows[i] % 16 != 0)) return false;
    return true;
}

int gemm_rows_mfma_launch(const GmArgs &P0, int epi, int norm, hipStream_t s)
{
    GmArgs P = P0;
    P.total_rows = 0;
    for (int i = 0; i < P.nsets; i++) P.total_rows += P.rows[i];
    if (!gemm_rows_mfma_fused_ok(P, epi, norm)) return ifa_fail(IFA_ERR_ARG, "rows GEMM: shape not covered (T %d, %d blocks, %d sets)", P.T, P.nblk, P.nsets);
    int wgs, maxt;
    gm_geometry((size_t)P.total_rows, gm_num_cus(), &wgs, &maxt);
    if (epi == GM_GLU) {                       // pairs of tiles: 1, 2, 3, 4 pairs per workgroup
        if (maxt > 4) { maxt = 4; wgs = ((P.total_rows + 15) / 16 + 3) / 4; }
        maxt = maxt == 3 ? 6 : maxt * 2;
    }
    const size_t smem = gm_smem(P.T, maxt);
    switch (maxt) {
    case 1: return gm_launch1<1>(wgs, smem, P, epi, norm, s);
    case 2: return gm_launch1<2>(wgs, smem, P, epi, norm, s);
    case 3: return gm_launch1<3>(wgs, smem, P, epi, norm, s);
    case 4: return gm_launch1<4>(wgs, smem, P, epi, norm, s);
    case 6: return gm_launch1<6>(wgs, smem, P, epi, norm, s);
    default: return gm_launch1<8>(wgs, smem, P, epi, norm, s);
    }
}

template <int MT>
static int gm_launch_grouped(int wgs, int groups, size_t smem, const MoeSmallGroup &grp, size_t rows, size_t cols, const void *X, void *Y, hipStream_t s)
{
    auto kern = k_gemm_rows_mfma_grouped<MT, 8>;
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)wgs, (unsigned)groups), dim3(GM_THREADS), smem, s>>>(grp, (int)rows, (int)(cols / 32), (const half_t *)X, (half_t *)Y);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int gemm_rows_mfma(const void *Wt_tiled, size_t rows, size_t cols, const void *x_f16, size_t tokens, const void *bias_f16, void *y_f16,
                   hipStream_t s)
{
    if (!gemm_rows_mfma_ok(rows, cols, tokens)) return IFA_ERR_STATE;
    GmArgs P; memset(&P, 0, sizeof(P));
    P.W[0] = (const uint8_t *)Wt_tiled; P.rows[0] = (int)rows; P.nsets = 1; P.nblk = (int)(cols / 32);
    P.T = (int)tokens;                       // 9..16 rows: activation chunks of 2048 columns (GmGeo<16>)
    P.X = (const half_t *)x_f16; P.ldx = (int)cols;
    P.bias[0] = (const half_t *)bias_f16;
    P.Y = (half_t *)y_f16; P.ldy = (int)rows;
    return gemm_rows_mfma_launch(P, GM_PLAIN, 0, s);
}

// rows / cols of ONE expert matrix; X / Y: the gathered activations / outputs of all entries; groups of 2..16 rows
int gemm_rows_mfma_grouped(const MoeSmallGroup &grp, size_t rows, size_t cols, const void *X, void *Y, int max_groups, int max_rows, hipStream_t s)
{
    if (!gemm_rows_mfma_ok(rows, cols, 2)) return ifa_fail(IFA_ERR_STATE, "grouped rows GEMM: %zu x %zu", rows, cols);
    if (max_groups <= 0) return IFA_OK;
    if (max_rows > 8) return ifa_fail(IFA_ERR_ARG, "grouped rows GEMM: groups of up to %d rows (limit 8)", max_rows);
    int wgs, maxt;
    gm_geometry(rows, std::max(32, 2 * gm_num_cus() / max_groups), &wgs, &maxt);     // the experts share the chip
    const size_t smem = gm_smem(8, maxt);
    switch (maxt) {
    case 1: return gm_launch_grouped<1>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    case 2: return gm_launch_grouped<2>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    case 3: return gm_launch_grouped<3>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    case 4: return gm_launch_grouped<4>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    case 6: return gm_launch_grouped<6>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    default: return gm_launch_grouped<8>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    }
}

} // namespace ifa
