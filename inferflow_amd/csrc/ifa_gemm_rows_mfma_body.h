// ifa_gemm_rows_mfma_body.h -- device body of the 2..16-row weight-streaming GEMM on the matrix cores, shared by its two
// translation units: ifa_gemm_rows_mfma.hip (weights in the tiled layout, launch interface, MoE groups) and
// ifa_gemm_rows_mo.hip (weights in the MO layout).  See ifa_gemm_rows_mfma.hip for the design notes.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include "ifa_host.h"
#include "ifa_decode_kernels.h"
#include "ifa_moe.h"
#include "ifa_gemm_rows_mfma.h"
#include "ifa_dequant_q4.h"

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
// MO: the weights are the MFMA-operand-order copy (ifa_gemm_rows_mfma.h "MO layout", built once per tensor by
// gemm_rows_mo_build): a lane's request IS its A operand -- block 4S + g of row r for lane (r, g) -- so a superstep of a tile is
// one contiguous KiB per wave and the LDS patch (store, wait, read back: a ~1.2 us dependent chain per group) disappears.
// Same operands, same MFMA order: bit-identical to the tiled path.
// CH: 1 = the K range is ONE chunk of activation rows (checked by the launcher): straight-line code with counted waits
//     2 = rows longer than a chunk staged WHOLE, once (MO, <= 4 rows, no norm: w2 of a batched step of 2..4 queries, the 2..4-row
//         expert groups of a MoE step): the weight groups still walk 4096-column chunks, but no chunk is re-staged -- the chunk
//         loop of CH = 0 stages, passes two barriers and drains its waves once per chunk (rows-trace at 2 queries, K = 11008:
//         the data is in by 5 us, the loop ends at 10)
#ifndef IFA_ROWS_BARRIER_FIRST
#define IFA_ROWS_BARRIER_FIRST 1
#endif
#ifndef IFA_ROWS_FETCH_EARLY
#define IFA_ROWS_FETCH_EARLY 1
#endif
// whole-row staging (CH == 2), order of the prologue: 0 = every weight group requested, then the rows stored (counted wait);
// 1 = ONE group in flight, rows stored, barrier, then the other groups; 2 = rows stored and the barrier passed before ANY
// weight request
#ifndef IFA_ROWS_WHOLE_ORDER
#define IFA_ROWS_WHOLE_ORDER 0
#endif
template <int MAXT, int TX, int EPI, int NORM, bool MO, int CH, int KPM = 0>
__device__ __forceinline__ void gemm_rows_mfma_body(const GmArgs &P, char *smem)
{
    static_assert(KPM == 0 || (CH == 0 && EPI != GM_GLU && NORM == 0 && MO && MAXT % KPM == 0), "K parts: chunk loop, MO layout, no gated pair, no norm prologue");
    // GM_GLU: a workgroup's tiles come in PAIRS -- `it` even: tile (it / 2) of w1, odd: the same tile of w3 -- so the gated
    // product keeps the plain kernel's registers and prefetch depth (MAXT counts both; the epilogue pairs the accumulators)
    constexpr bool GLU = EPI == GM_GLU;
    static_assert(!GLU || MAXT % 2 == 0, "GM_GLU: tiles per workgroup come in pairs");
    // MO: no patches, so 16 rows x 4096 columns fit too (131 KB); 17..32 rows (MO only): TWO 16-query column tiles share every
    // dequantised A operand (NT = 2), staged in 2048-column chunks (32 rows x 4 KB)
    constexpr int NT = TX > 16 ? 2 : 1;
    static_assert(NT == 1 || (MO && NORM == 0 && CH == 0), "17..32 rows: MO layout, chunked rows, norm as its own launch");
    constexpr bool WHOLE = CH == 2;
    static_assert(!WHOLE || (MO && NORM == 0 && TX <= 4), "whole-row staging: MO layout, <= 4 rows, no norm prologue");
    using G = GmGeo<(MO && TX <= 16) ? 32 : gm_cs(TX)>;
    using GmGrp = GmGrpT<G::NI>;
    static_assert(NORM == 0 || G::CHUNK_SUP == 32, "the norm prologue needs the whole row in one chunk");
    constexpr int PD = (MO && G::NJ == 2) ? 5 : 3;    // groups in flight per wave (a group is 5 KB, or 2.5 KB with 2048-column chunks)
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
    const int nq4 = (nsup + 3) >> 2;                              // MO: header quads per tile
    const size_t mo_tile = (size_t)(nsup + nq4) * 1024;           // MO: bytes per 16-row tile
    const float fp8_up = q4_fp8_up();                             // (ifa_dequant_q4.h)
    const int row_stride = WHOLE ? K * 2 + 16 : G::ROW_STRIDE;    // bytes per activation row in LDS
    const int trow = min(r, T - 1);                               // B operand: token of this lane (columns past T: duplicates, never stored)
    const int trow1 = min(r + 16, T - 1);                         // second column tile (NT == 2)

    // a group = this wave's 16 rows x BPW blocks of (tile, chunk): blocks blk0 .. blk0 + BPW - 1, blk0 = 4 CS chunk + BPW wave
    // valid == false (past the last group): all lanes re-read the first bytes of the matrix -- one cache line, no branch
    // K parts (KPM > 0, GmArgs::kparts): workgroup ids kparts * g + kp share tile group g
    const int kparts = KPM > 0 ? MAXT / KPM : 1;
    const int kp = KPM > 0 ? (int)blockIdx.x % kparts : 0;
    const int wgx = KPM > 0 ? (int)blockIdx.x / kparts : (int)blockIdx.x, nwgx = KPM > 0 ? (int)gridDim.x / kparts : (int)gridDim.x;
    auto tile_of = [&](int it) { return GLU ? wgx + (it >> 1) * nwgx : wgx + it * nwgx; };
    constexpr int RPI = 64 / G::BPW;               // rows per code request
    constexpr int LPR = G::BPW / 4;                // lanes per row of (base, scale) words (4 blocks' words each)
    // Weight requests are GLOBAL loads through an explicit address-space cast: the set pointers pass through integers
    // (readfirstlane, gm_sel), so the compiler no longer knows their address space and would emit FLAT loads -- which count
    // on lgkmcnt as well, so that every LDS wait in the compute phase (lgkmcnt(0)) also waited for all weight groups in flight:
    // the kernel requested everything, waited, then computed (rows-trace timeline, DESIGN.md "Dynamic batching").
    typedef const __attribute__((address_space(1))) u32x4 gu4;
    auto gload = [](const uint8_t *base, uint32_t off) { return __builtin_nontemporal_load((gu4 *)(base + off)); };
    auto fetch = [&](GmGrp &q, int it, int chunk, bool valid) {
        const GmTile tl = gm_locate(S, min(tile_of(it), ntiles - 1));
        const int blk0 = chunk * (G::CHUNK_SUP * 4) + wave * G::BPW;
        const uint8_t *Wt = gm_sel(GLU && (it & 1), W1p, tl.W0);       // (by value: see gm_sel)
        if constexpr (MO) {
            // supersteps S0 .. S0 + NJ - 1 of tile row0 / 16: one KiB each, lane l at 16 l; their (base, scale) words: quad
            // S0 / 4, lane l at 16 l (word j = superstep 4 (S0 / 4) + j).  32-bit offsets (a matrix is < 4 GB); past the last
            // group every wave re-reads the first KiB of the set (scalar selects on cheap values: no branch around a load)
            const int S0 = chunk * G::CHUNK_SUP + wave * G::NJ;
            const uint32_t tile_off = (uint32_t)(tl.row0 >> 4) * (uint32_t)mo_tile;
            const uint32_t lo = (uint32_t)lane * 16u;
#pragma unroll
            for (int j = 0; j < G::NJ; j++) {
                const uint32_t o = tile_off + (uint32_t)min(S0 + j, nsup - 1) * 1024u;
                q.c[j] = gload(Wt, (valid ? o : 0u) + lo);
            }
            const uint32_t oq = tile_off + (uint32_t)(nsup + min(S0 >> 2, nq4 - 1)) * 1024u;
            q.sb = gload(Wt, (valid ? oq : 0u) + lo);
        } else {
            const uint32_t rb = (uint32_t)row_bytes;
#pragma unroll
            for (int i = 0; i < G::NI; i++) {      // codes: lane l -> row RPI i + l / BPW, block l % BPW (BPW * 16 contiguous bytes per row)
                const int row = valid ? min(tl.row0 + RPI * i + lane / G::BPW, tl.nrows - 1) : 0;
                const int blk = valid ? min(blk0 + (lane % G::BPW), nblk - 1) : 0;
                q.c[i] = gload(Wt, (uint32_t)row * rb + (uint32_t)blk * 16u);
            }
            {                                       // (base, scale): 16 rows x LPR lanes, four blocks' words per lane (upper lanes: duplicates)
                const int ls = lane & (16 * LPR - 1);
                const int row = valid ? min(tl.row0 + ls / LPR, tl.nrows - 1) : 0;
                const int blk = valid ? min(blk0 + 4 * (ls % LPR), nblk - 4) : 0;
                q.sb = gload(Wt, (uint32_t)row * rb + (uint32_t)nblk * 16u + (uint32_t)blk * 4u);
            }
        }
    };
    char *patch = smem + (size_t)TX * G::ROW_STRIDE + (size_t)wave * G::PATCH_BYTES;        // this wave's transposition patch
    auto compute = [&](const GmGrp &q, int chunk, f4m &acc, f4m &acc1) {
        if constexpr (!MO) {
        // ---- through the patch: rows of BPW code blocks and rows of BPW (base, scale) words, both at a stride that makes the
        // 16-row reads below conflict-free
#pragma unroll
        for (int i = 0; i < G::NI; i++)
            *reinterpret_cast<u32x4 *>(patch + (size_t)(RPI * i + lane / G::BPW) * G::CSTRIDE + (size_t)(lane % G::BPW) * 16) = q.c[i];
        if (lane < 16 * LPR) *reinterpret_cast<u32x4 *>(patch + 16 * G::CSTRIDE + (size_t)(lane / LPR) * G::SSTRIDE + (size_t)(lane % LPR) * 16) = q.sb;
        }
        const int blk0 = chunk * (G::CHUNK_SUP * 4) + wave * G::BPW;
        const bool upper = MO && G::NJ == 2 && ((blk0 >> 2) & 2);      // MO, two supersteps per group: words 2, 3 of the quad
#pragma unroll
        for (int j = 0; j < G::NJ; j++) {
            if (blk0 + 4 * j >= nblk) continue;                    // wave-uniform: past the row end (nblk % 4 == 0)
            u32x4 cw4;
            uint32_t sbw;
            if constexpr (MO) {
                cw4 = q.c[j];
                if constexpr (G::NJ == 2) sbw = upper ? q.sb[2 + j] : q.sb[j];
                else sbw = q.sb[j];
            } else {
                cw4 = *reinterpret_cast<const u32x4 *>(patch + (size_t)r * G::CSTRIDE + (size_t)(4 * j + g) * 16);
                sbw = *reinterpret_cast<const uint32_t *>(patch + 16 * G::CSTRIDE + (size_t)r * G::SSTRIDE + (size_t)(4 * j + g) * 4);
            }
            const float base = hbits2f((uint16_t)(sbw & 0xFFFFu)), scale_up = hbits2f((uint16_t)(sbw >> 16)) * fp8_up;
            const int xcol = (WHOLE ? chunk * G::CHUNK_COLS : 0) + (wave * G::BPW + 4 * j + g) * 32;
            const char *xrow = smem + (size_t)trow * row_stride + (size_t)xcol * 2;
            const char *xrow1 = smem + (size_t)trow1 * row_stride + (size_t)xcol * 2;
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                // byte b of the word: low nibble = element 8 s4 + 2b, high nibble = the next one (ifa_gemm_rows.hip); the reference's
                // dequantised halves, two codes per conversion (ifa_dequant_q4.h: 15 VALU instructions per 8 weights, was 19)
                q4_h2 wq[4];
                q4x8_dequant(cw4[s4], scale_up, base, wq);
                const h8m a = {wq[0][0], wq[0][1], wq[1][0], wq[1][1], wq[2][0], wq[2][1], wq[3][0], wq[3][1]};
                const h8m b = *reinterpret_cast<const h8m *>(xrow + s4 * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
                if constexpr (NT == 2) {
                    const h8m b1 = *reinterpret_cast<const h8m *>(xrow1 + s4 * 16);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc1, 0, 0, 0);
                }
            }
        }
    };

    f4m acc[MAXT], acc1[NT == 2 ? MAXT : 1];
#pragma unroll
    for (int i = 0; i < MAXT; i++) acc[i] = f4m{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < (NT == 2 ? MAXT : 1); i++) acc1[i] = f4m{0.0f, 0.0f, 0.0f, 0.0f};
    // PD groups of requests in flight per wave (a group = 4 supersteps = 5 KB per wave): one group ahead left every wave
    // waiting ~half of the time for HBM
    GmGrp buf[PD];
    // this workgroup's chunks [c_lo, c_hi) of K (all of them without K parts; the launcher makes nchunk == kparts * cpp)
    const int cpp = nchunk / kparts, c_lo = kp * cpp, c_hi = c_lo + cpp;
    const int nq = cpp * MAXT;                                     // (chunk, tile) pairs in execution order: chunk outer
    // (one chunk: the groups past the last one are not requested at all -- qi is a literal after unrolling, the test folds away.
    //  As clamped dummy loads they were 5 of the 8 group requests of a wave of the wq | wk | wv launch: ~0.6 us of the CU's
    //  request path.  The chunk loop keeps them: a branch around a load makes every later wait a vmcnt(0).)
    auto fetch_q = [&](GmGrp &q, int qi) __attribute__((always_inline)) {
        if constexpr (CH == 1) { if (qi >= MAXT) return; }
        const int ch = qi / MAXT; fetch(q, qi - ch * MAXT, c_lo + ch, qi < nq);
    };
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
    // whole rows (CH == 2): piece tid + 512 p of every row, p < 4 (K <= 16384); pieces past the row end are duplicates of its last
    // piece -- requested and stored unconditionally (same bytes to the same place), like the clamped rows above
    constexpr int WP = WHOLE ? 4 : 1;
    u32x4 xw[WHOLE ? TX : 1][WP];
    auto xw_request = [&]() {
        const int pieces = K >> 3;
#pragma unroll
        for (int k = 0; k < (WHOLE ? TX : 1); k++)
#pragma unroll
            for (int pp = 0; pp < WP; pp++)
                xw[k][pp] = *reinterpret_cast<const u32x4 *>(Xp + (size_t)min(k, T - 1) * ldx + (size_t)min(tid + GM_THREADS * pp, pieces - 1) * 8);
    };
    auto xw_store = [&]() {
        const int pieces = K >> 3;
#pragma unroll
        for (int k = 0; k < (WHOLE ? TX : 1); k++)
#pragma unroll
            for (int pp = 0; pp < WP; pp++)
                *reinterpret_cast<u32x4 *>(smem + (size_t)k * row_stride + (size_t)min(tid + GM_THREADS * pp, pieces - 1) * 16) = xw[k][pp];
    };
    long long *const trc_x = P.trace ? P.trace + (size_t)blockIdx.x * 32 : nullptr;
    auto x_store = [&](int chunk, bool first_chunk = false) {
        const int per_row = min(G::CHUNK_COLS, K - chunk * G::CHUNK_COLS) >> 3;
        if constexpr (NORM == 1) {
            // RMS norm of every row in the canonical order of ifa_math.h: piece c = tid is lane c % 64 of group c / 64 = wave
            // (NORM variants stage one row per pass: XR == 1)
            float *part = reinterpret_cast<float *>(smem + (size_t)TX * G::ROW_STRIDE + (MO ? (size_t)0 : (size_t)GM_WAVES * G::PATCH_BYTES));     // [TX][8]
            // wave_sum() of every row, stage by stage ACROSS the rows (same operations per row, so the same sums): row after row, its
            // two LDS permutes (lanes ^ 16, ^ 32) were 16 dependent round trips per wave
            float pg[XK];
#pragma unroll
            for (int k = 0; k < XK; k++) {
                rms_h8 v8 = __builtin_bit_cast(rms_h8, xv[k]);
                if (tid >= per_row) {
#pragma unroll
                    for (int i = 0; i < 8; i++) v8[i] = (half_t)0;
                }
                pg[k] = rms_chunk_sq(v8);
            }
#pragma unroll
            for (int k = 0; k < XK; k++) pg[k] += dpp_xor1(pg[k]);
#pragma unroll
            for (int k = 0; k < XK; k++) pg[k] += dpp_xor2(pg[k]);
#pragma unroll
            for (int k = 0; k < XK; k++) pg[k] += dpp_half_mirror(pg[k]);
#pragma unroll
            for (int k = 0; k < XK; k++) pg[k] += dpp_mirror(pg[k]);
            float pt[XK];
#pragma unroll
            for (int k = 0; k < XK; k++) pt[k] = __shfl_xor(pg[k], 16);
#pragma unroll
            for (int k = 0; k < XK; k++) pg[k] += pt[k];
#pragma unroll
            for (int k = 0; k < XK; k++) pt[k] = __shfl_xor(pg[k], 32);
#pragma unroll
            for (int k = 0; k < XK; k++) pg[k] += pt[k];
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < XK; k++) part[k * GM_WAVES + wave] = pg[k];
            }
            if (trc_x && tid == 0) trc_x[6] = wall_clock64();
            __syncthreads();
            if (trc_x && tid == 0) trc_x[7] = wall_clock64();
            const rms_h8 nw8 = __builtin_bit_cast(rms_h8, nwv);
            // Every workgroup normalises all T rows (each needs the whole image): the VALU work per thread is what this costs
            // (2.2 us when every thread derived all eight row scales with IEEE division and square root: rows-trace).  Lane l derives
            // the scale of row l % XK only -- the eight group sums of that row in ascending order (groups past the row end hold +0:
            // exact), read as two 16-byte words -- and the rows' scales are read back lane by lane (v_readlane); the element-wise
            // part runs on packed fp32 multiplies.  Same operations per element as rms_apply (ifa_math.h).
            const int myrow = lane & (XK - 1);
            const f4m pa = *reinterpret_cast<const f4m *>(part + myrow * GM_WAVES), pb = *reinterpret_cast<const f4m *>(part + myrow * GM_WAVES + 4);
            float total = 0.0f;
            total = total + pa[0]; total = total + pa[1]; total = total + pa[2]; total = total + pa[3];
            total = total + pb[0]; total = total + pb[1]; total = total + pb[2]; total = total + pb[3];
            const float my_scale = rms_scale_of(total, K, P.eps);
            f2m mlt[4];                                   // base + weight per element (1 when there is no weight: x * 1 is exact)
#pragma unroll
            for (int i = 0; i < 4; i++)
                mlt[i] = nwp ? f2m{P.multi_base + (float)nw8[2 * i], P.multi_base + (float)nw8[2 * i + 1]} : f2m{1.0f, 1.0f};
#pragma unroll
            for (int k = 0; k < XK; k++) {
                const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_scale), k));
                const f2m sc2 = {scale, scale};
                const rms_h8 v8 = __builtin_bit_cast(rms_h8, xv[k]);
                rms_h8 o;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    f2m t = f2m{(float)v8[2 * i], (float)v8[2 * i + 1]} * sc2;
                    t = t * mlt[i];
                    const h2m h = __builtin_convertvector(t, h2m);
                    o[2 * i] = h[0]; o[2 * i + 1] = h[1];
                }
                xv[k] = __builtin_bit_cast(u32x4, o);
            }
        }
#if IFA_ROWS_FETCH_EARLY
        if constexpr (NORM == 0) {
            // Round 4: plain staging (wo, w2, all 17..32-row launches): the other PD - 1 groups are requested BEFORE the rows are
            // stored -- behind the rows' own requests in this CU's queue, so the rows still return first -- instead of behind the
            // staging barrier: 32 queries 3.69 -> 3.36 ms per step, 16 queries 2.28 -> 2.18 (2 queries: unchanged)
            if (first_chunk) {
#pragma unroll
                for (int d = 1; d < PD; d++) fetch_q(buf[d], d);
                if (trc_x && tid == 0) trc_x[18] = wall_clock64();
            }
        }
#endif
        if (trc_x && tid == 0) trc_x[16] = wall_clock64();
        // UNCONDITIONAL stores (rows past T hold duplicates of row T - 1, pieces past the row end duplicates of its last piece: both
        // inside the image, never read as data): a branch around the store made the wait in front of it vmcnt(0), i.e. the staging
        // waited for every weight group already in flight
#pragma unroll
        for (int k = 0; k < XK; k++) {
            const int row = xsub + G::XR * k;
            *reinterpret_cast<u32x4 *>(smem + (size_t)row * G::ROW_STRIDE + (size_t)xpiece * 16) = xv[k];
        }
        if (trc_x && tid == 0) trc_x[17] = wall_clock64();
    };
    long long *const trc = P.trace ? P.trace + (size_t)blockIdx.x * 32 : nullptr;
    if (trc && tid == 0) trc[0] = wall_clock64();
    if constexpr (WHOLE) xw_request(); else x_request(c_lo);
    __syncthreads();
    if (trc && tid == 0) trc[1] = wall_clock64();
    // ONE group per wave is requested in front of the staging, the rest behind it: with all PD groups (30 MB chip-wide for a
    // 12288-row matrix) queued first, the rows of late-starting workgroups sat behind them in the memory system -- staged at
    // 4.4 us instead of 1.4 (rows-trace)
    if constexpr (!(WHOLE && IFA_ROWS_WHOLE_ORDER == 2)) fetch_q(buf[0], 0);
    // The waits below must be COUNTED waits: the first chunk's rows are the oldest requests (vmcnt = the weight loads issued
    // after them), so staging them does not wait for the weight groups in flight.  That needs straight-line code: the
    // single-chunk case (K <= 4096: wq | wk | wv, wo, w1 / w3) is its own path, and in the chunk loop the NEXT chunk's rows
    // are requested before the current chunk's weight groups and stored after them.
    auto run_chunk = [&](int chunk) __attribute__((always_inline)) {           // chunk: absolute; the groups are counted from c_lo
#pragma unroll
        for (int it = 0; it < MAXT; it++) {
            const int qi = (chunk - c_lo) * MAXT + it;
            // The two waves of a SIMD (w, w + 4) are both VALU-ready most of the time and the issue arbiter favours the older one:
            // waves 0..3 ran ahead (all their groups done while waves 4..7 had finished two of eight: rows-trace), i.e. only half of
            // the CU's requests were cycling.  Alternate the priority per group so the pair takes turns.
            if (((qi + (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
            compute(buf[0], chunk, acc[it], acc1[NT == 2 ? it : 0]);
            if (trc && tid == 0 && qi == 0) trc[3] = wall_clock64();
#pragma unroll
            for (int d = 0; d + 1 < PD; d++) buf[d] = buf[d + 1];
            fetch_q(buf[PD - 1], qi + PD);
        }
    };
    if constexpr (WHOLE) {
        if constexpr (IFA_ROWS_WHOLE_ORDER == 0) {
#pragma unroll
            for (int d = 1; d < PD; d++) fetch_q(buf[d], d);      // (behind the rows' own requests in this CU's queue, see x_store)
        }
        xw_store();
    } else x_store(c_lo, true);
    // The staging barrier FIRST, the other PD - 1 groups behind it (round 4): requested in front of the barrier, every wave sat in
    // its load issue (4 x 5 KB per wave against a full memory queue) before it could arrive -- the rows were staged at 3.0 us and
    // the barrier passed at 4.8 (rows-trace: "other groups requested 3.22, x staged 4.77"); the first group is in flight since 0.8
#if IFA_ROWS_FETCH_EARLY
    __syncthreads();
    if (trc && tid == 0) trc[2] = wall_clock64();
    if constexpr (WHOLE && IFA_ROWS_WHOLE_ORDER != 0) {
#pragma unroll
        for (int d = (IFA_ROWS_WHOLE_ORDER == 2 ? 0 : 1); d < PD; d++) fetch_q(buf[d], d);
        if (trc && tid == 0) trc[18] = wall_clock64();
    }
    if constexpr (NORM == 1) {      // (norm prologue: requested from inside the staging, every wave would sit in its load issue for ~1.2 us with the rows still to scale)
#pragma unroll
        for (int d = 1; d < PD; d++) fetch_q(buf[d], d);
        if (trc && tid == 0) trc[18] = wall_clock64();
    }
#elif IFA_ROWS_BARRIER_FIRST
    __syncthreads();
    if (trc && tid == 0) trc[2] = wall_clock64();
#pragma unroll
    for (int d = 1; d < PD; d++) fetch_q(buf[d], d);
    if (trc && tid == 0) trc[18] = wall_clock64();
#else
#pragma unroll
    for (int d = 1; d < PD; d++) fetch_q(buf[d], d);
    if (trc && tid == 0) trc[18] = wall_clock64();
    __syncthreads();
    if (trc && tid == 0) trc[2] = wall_clock64();
#endif
    if constexpr (CH == 1) {
        run_chunk(0);
    } else if constexpr (WHOLE) {
        for (int chunk = 0; chunk < nchunk; chunk++) run_chunk(chunk);
    } else {
        for (int chunk = c_lo; chunk < c_hi; chunk++) {
            const int nx = min(chunk + 1, c_hi - 1);
            if (cpp > 1) x_request(nx);                            // (last chunk: re-read, never stored)
            run_chunk(chunk);
            if (chunk + 1 < c_hi) {
                __syncthreads();                                   // this chunk's fragments have been read
                x_store(nx);
                __syncthreads();
            }
        }
    }
    // ---- sum the 8 waves' partial tiles in wave order, then the epilogue: thread e of the first 256 owns element
    // (m = (l >> 4) * 4 + i, n = l & 15) of every tile, l = e >> 2, i = e & 3 (the MFMA's C layout)
    if (trc && lane == 0) trc[8 + wave] = wall_clock64();
    __syncthreads();                                               // the activation image is free: partials take its place
    if (trc && tid == 0) trc[4] = wall_clock64();
    float *part = reinterpret_cast<float *>(smem);                 // [NT][MAXT][8 waves][256]
#pragma unroll
    for (int it = 0; it < MAXT; it++) {
        *reinterpret_cast<f4m *>(part + ((size_t)(it * GM_WAVES + wave) * 64 + lane) * 4) = acc[it];
        if constexpr (NT == 2) *reinterpret_cast<f4m *>(part + ((size_t)((MAXT + it) * GM_WAVES + wave) * 64 + lane) * 4) = acc1[it];
    }
    __syncthreads();
    {
        // both halves of the workgroup: thread e = tid & 255 owns element e of the tiles (pairs) whose index parity is tid >> 8
        const int e = tid & 255, hsel = tid >> 8;
        const int l = e >> 2, i = e & 3;
        const int m = (l >> 4) * 4 + i;
        auto total = [&](int it) {
            float sum = 0.0f;
#pragma unroll
            for (int w = 0; w < GM_WAVES; w++) sum = sum + part[(size_t)(it * GM_WAVES + w) * 256 + e];
            return sum;
        };
        constexpr int STEP = GLU ? 2 : 1;
        // bias / residual / gate of one output element (tile vt, column tile nt) and its store
        auto emit = [&](int vt, int nt, float s0, float s1) {
            const int n = (l & 15) + 16 * nt;
            const GmTile tl = gm_locate(S, min(vt, ntiles - 1));
            const int row = tl.row0 + m;
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
        };
        if constexpr (KPM > 0) {
            // ---- K parts, reduce-scatter form.  (One finishing workgroup per group was prototyped first: its CU pulled 123 KB of
            // partial sums at ~25 GB/s while the others idled -- 8-11 us behind a loop that had gone from 21 to 11 us.)
            unsigned long long *const gsum = P.kpart_sums;
            auto slot = [&](int vt, int part_id, int nt) { return gsum + (((size_t)vt * kparts + part_id) * NT + nt) * 256 + e; };
            // A. the sums of the tiles other workgroups finish: write-through granules, nothing waits for them here
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int it = 0; it < MAXT; it++) {
                    if ((it & 1) != hsel || it / KPM == kp) continue;
                    const int vt = tile_of(it);
                    if (vt >= ntiles) continue;
                    const float v = total(nt * MAXT + it);
                    __hip_atomic_store(slot(vt, kp, nt), (1ull << 32) | (unsigned long long)__builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            // B. the tiles this workgroup finishes: its own part from LDS, the other parts' granules polled (all requests first)
            float own[NT][KPM];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int j = 0; j < KPM; j++) {
                    const int it = kp * KPM + j;
                    float sum = 0.0f;
#pragma unroll
                    for (int w = 0; w < GM_WAVES; w++) sum = sum + part[(size_t)((nt * MAXT + it) * GM_WAVES + w) * 256 + e];
                    own[nt][j] = sum;
                }
            unsigned long long u[NT][KPM][7];
            const long long t_wait = wall_clock64();
            for (;;) {
                bool all = true;
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int j = 0; j < KPM; j++) {
                        const int it = kp * KPM + j;
                        if ((it & 1) != hsel) continue;         // (wave-uniform: the other half of the workgroup finishes the tiles of the other parity)
                        const int vt = min(tile_of(it), ntiles - 1);
#pragma unroll
                        for (int z = 0; z < 7; z++) {           // the other parts in K order: z < kp -> part z, else part z + 1 (clamped duplicates past the end: never looked at)
                            const int q = min(z < kp ? z : z + 1, kparts - 1);
                            u[nt][j][z] = __hip_atomic_load(slot(vt, q, nt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int j = 0; j < KPM; j++) {
                        const int it = kp * KPM + j;
                        if ((it & 1) != hsel || tile_of(it) >= ntiles) continue;
#pragma unroll
                        for (int z = 0; z < 7; z++) if (z < kparts - 1) all = all && (u[nt][j][z] >> 32) == 1ull;
                    }
                if (__builtin_amdgcn_ballot_w64(!all) == 0ull) break;           // (per wave: every lane's granules are in)
                if (wall_clock64() - t_wait > WAIT_TIMEOUT_TICKS) {              // the partners are not resident: leave a code and go on (the host fails the call
                    if (P.wait_err && (tid & 63) == 0) __hip_atomic_store(P.wait_err, 0x71u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // and switches the waiting launches off)
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int j = 0; j < KPM; j++) {
                    const int it = kp * KPM + j;
                    if ((it & 1) != hsel) continue;
                    const int vt = tile_of(it);
                    if (vt >= ntiles) continue;
                    // K order: parts 0 .. kp - 1, own, kp + 1 .. kparts - 1
                    float sum = 0.0f;
                    bool first = true;
#pragma unroll
                    for (int z = 0; z < 8; z++) {
                        if (z >= kparts) continue;
                        const float term = z == kp ? own[nt][j] : __builtin_bit_cast(float, (uint32_t)u[nt][j][z < kp ? z : z - 1]);
                        sum = first ? term : sum + term;
                        first = false;
                    }
#pragma unroll
                    for (int z = 0; z < 8; z++)
                        if (z < kparts && z != kp) __hip_atomic_store(slot(vt, z, nt), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                    emit(vt, nt, sum, 0.0f);
                }
            if (trc && tid == 0) trc[5] = wall_clock64();
            return;
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int n = (l & 15) + 16 * nt;
#pragma unroll
            for (int it = 0; it < MAXT; it += STEP) {
                if (((it / STEP) & 1) != hsel) continue;
                const int vt = tile_of(it);
                const GmTile tl = gm_locate(S, min(vt, ntiles - 1));
                const int row = tl.row0 + m;
                const float s0 = total(nt * MAXT + it);
                float s1 = 0.0f;
                if constexpr (GLU) s1 = total(nt * MAXT + it + 1);
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
    if (trc && tid == 0) trc[5] = wall_clock64();
}

template <int MAXT, int TX, int EPI, int NORM, bool MO, int CH, int KPM = 0>
__global__ void __launch_bounds__(GM_THREADS) k_gemm_rows_mfma(const GmArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_rows_mfma_body<MAXT, TX, EPI, NORM, MO, CH, KPM>(P, smem);
}

static int gm_tx(int T) { return T <= 2 ? 2 : (T <= 4 ? 4 : (T <= 8 ? 8 : 16)); }
// whole-row staging (CH == 2): tx rows of K columns
static size_t gm_smem_whole(int tx, int K, int maxt) { return std::max((size_t)tx * ((size_t)K * 2 + 16), (size_t)maxt * GM_WAVES * 256 * 4); }
static size_t gm_smem(int T, int maxt, int mo)
{
    const int tx = mo ? (T <= 2 ? 2 : (T <= 4 ? 4 : (T <= 8 ? 8 : (T <= 16 ? 16 : 32)))) : gm_tx(T);
    if (mo && tx == 32) return std::max((size_t)32 * GmGeo<16>::ROW_STRIDE, (size_t)2 * maxt * GM_WAVES * 256 * 4);
    if (mo) return std::max((size_t)tx * GmGeo<32>::ROW_STRIDE + (size_t)16 * GM_WAVES * 4, (size_t)maxt * GM_WAVES * 256 * 4);
    const size_t row = tx > 8 ? GmGeo<16>::ROW_STRIDE : GmGeo<32>::ROW_STRIDE, patch = tx > 8 ? GmGeo<16>::PATCH_BYTES : GmGeo<32>::PATCH_BYTES;
    const size_t ximg = (size_t)tx * row + (size_t)GM_WAVES * patch + 8 * GM_WAVES * 4;      // + the norm's group sums
    const size_t parts = (size_t)maxt * GM_WAVES * 256 * 4;
    return std::max(ximg, parts);
}

// MO-layout kernels (ifa_gemm_rows_mo.hip); wgs / maxt from gemm_rows_mfma_launch's geometry
int gemm_rows_mo_launch(const GmArgs &P, int epi, int norm, int wgs, int maxt, hipStream_t s);
// the experts with 2..8 rows of a mixture-of-experts layer on their MO copies; glu: w1 / w3 pairs with act(.) * (.) as the output
int gemm_rows_mo_grouped(const MoeSmallGroup &grp, size_t rows, size_t cols, const void *X, void *Y, int max_groups, int glu, int act_kind, hipStream_t s);

} // namespace ifa
