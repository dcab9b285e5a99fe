// ifa_gemm_mid.hip -- the linear layers of a 33..512-token prompt:  Y[T][N] = X[T][K] . W[N][K]^T  (+ bias, residual, GLU)
//
// Reference: MatrixMultiplication (src/transformer/inference_worker.cc:2364-2432): Dequantize the weight tensor to F16, F16 x F16
// product with fp32 accumulation (src/tensor/cublas_engine.cu:420-436).  Same values here: every weight is half(fma(q, scale,
// base)) -- the reference's dequantised half -- products accumulate in fp32 in ascending 16-column groups of this workgroup's K
// range, parts of K added in order, one F16 rounding, bias / residual / gate as half operations (k_gemm_big's arithmetic).
//
// Why a second kernel (round 6).  k_gemm_big stages the activation tile with direct-to-LDS loads and reads its MFMA operands with
// ordinary LDS loads; hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS read that follows a direct-to-LDS request, so every
// 64-column step ends up waiting for the requests it has just issued: a step is one memory round trip (~1 us; rocprofv3 at 128
// tokens: W1|W3 62 us for 64 steps, wq|wk|wv 35 us for 16) however little it multiplies.  Above ~512 tokens the step's MFMA work
// covers that; below, a layer is 225 us for 126 MB of weights and 52 GFLOP.  Here:
//   * EVERYTHING global goes through direct-to-LDS loads into a ring of NS stages (activations 128 x 64 halves + the RAW 20-byte
//     weight blocks of the step: 21 KB per stage), requested NS - 1 steps ahead, waited for with a COUNTED vmcnt;
//   * every LDS read of the loop is inline assembly (the compiler sees no LDS access to protect), waited for with lgkmcnt where
//     the values are used;
//   * a wave owns 32 weight rows of the tile and ALL 128 tokens (4 accumulator tiles): it dequantises its rows' codes from the raw
//     bytes straight into the B operand registers (15 VALU per 8 weights, ifa_dequant_q4.h) -- no dequantised tile in LDS, no
//     second barrier -- while the MFMAs of the previous 16-column group run;
//   * one barrier per step (the activation tile is shared by the four waves); parts of K as in k_gemm_big.
// Weights: the MO copy of the rows GEMM (ifa_gemm_rows_mfma.h; Q4_B32T1A / B and the 64-weight nibble formats): a 16-row tile's K range
// is one linear stream of 1 KiB supersteps.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>
#include "ifa_host.h"
#include "ifa_math.h"
#include "ifa_tiled.h"
#include "ifa_gemm_big.h"
#include "ifa_dequant_q4.h"

namespace ifa {

typedef _Float16 md_h8 __attribute__((ext_vector_type(8)));
typedef float md_f16v __attribute__((ext_vector_type(16)));
typedef uint32_t md_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t md_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void md_lds_t;
typedef const __attribute__((address_space(1))) void md_glb_t;

#ifndef IFA_MID_LOADERS
#define IFA_MID_LOADERS 1   // 1: four more waves (one per SIMD) issue the direct-to-LDS requests; 0: the computing waves do, between their MFMAs
#endif
constexpr int MD_BN = 128, MD_BK = 64, MD_NW = 4, MD_NT = 256;      // (token rows of a tile: 32 TA, template parameter)
constexpr int md_a_bytes(int ta) { return 32 * ta * MD_BK * 2; }       // activation tile of a step (TA = 4: 16 KB), rows of 128 bytes, 16-byte chunks swizzled
constexpr int MD_WC_BYTES = MD_NW * 1024, MD_WS_BYTES = MD_NW * 256; // raw codes (32 rows x 2 blocks x 16 B per wave), (base, scale) words
constexpr int md_stage(int ta) { return md_a_bytes(ta) + MD_WC_BYTES + MD_WS_BYTES; }    // 21 KB (TA = 4) / 37 KB (TA = 8)
constexpr int MD_THREADS = IFA_MID_LOADERS ? 2 * MD_NT : MD_NT;     // launched threads (MD_NT of them compute and run the tails)
// measurement builds (-DIFA_MID_ABL=n): 1 = no activation requests, 2 = no MFMA, 3 = no dequantisation, 4 = no barrier, 5 = no weight requests
#ifndef IFA_MID_ABL
#define IFA_MID_ABL 0
#endif
#ifndef IFA_MID_PF
#define IFA_MID_PF 2        // 16-column groups of lead of the activation fragment reads (1: the first form, one group)
#endif
#ifndef IFA_MID_WAUX
#define IFA_MID_WAUX 0       // cache policy of the weight requests (2 = non-temporal)
#endif
constexpr int md_vm(int ta) { return IFA_MID_ABL == 1 ? 2 : (IFA_MID_ABL == 5 ? ta : ta + 2); }      // direct-to-LDS requests per wave and stage: TA (activations) + 2 (weights)

__device__ __forceinline__ md_u4 md_lds_b128(uint32_t addr) { md_u4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); return v; }
__device__ __forceinline__ uint32_t md_lds_b32(uint32_t addr) { uint32_t v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); return v; }

// EPI: GM_PLAIN | GM_RESIDUAL | GM_GLU (a weight tile = 64 rows of w1 and the same 64 rows of w3).  KS: parts of K (workgroups per
// tile).  NS: stages of the ring.  TA: 32-token accumulator tiles of a wave -- the tile is 32 TA tokens x 128 weight rows (TA = 8 from 512
// tokens on: every dequantised weight and every request of the weight stream serves twice the products).
template <int EPI, int KS, int NS, int TA>
__global__ void __launch_bounds__(MD_THREADS) k_gemm_mid(const GmArgs P, const BigGeo G)
{
    constexpr int BM = 32 * TA, BN = MD_BN, BK = MD_BK, NW = MD_NW, NT = MD_NT, ROWB = BK * 2;
    constexpr int MD_A_BYTES = md_a_bytes(TA), MD_STAGE = md_stage(TA), MD_VM = md_vm(TA), XP = TA;      // XP: activation pieces (8 token rows) a wave requests per stage
    constexpr bool GLU = EPI == GM_GLU;
    constexpr int BNE = GLU ? BN / 2 : BN;
    auto swz = [](int r) { return (r >> 1) & 7; };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long *const trc = (P.trace && threadIdx.x == 0) ? P.trace + (size_t)blockIdx.x * 8 : nullptr;      // (measurement: IFA_MID_TRACE=1)
    if (trc) trc[0] = wall_clock64();
    const int T = P.T, nblk = P.nblk, tiles_m = G.tiles_m;
    const half_t *__restrict__ X = P.X;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all & (NW - 1);              // a loader wave (IFA_MID_LOADERS) requests the pieces of the computing wave on its SIMD
    const bool loader = wave_all >= NW;
    // XCD-aware tile order, bands of token tiles (k_gemm_big's mapping)
    int wg;
    {
        const int nwg = (int)gridDim.x / KS, orig = (int)blockIdx.x % nwg, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int kz = KS == 1 ? 0 : (int)blockIdx.x / ((int)gridDim.x / KS);
    constexpr int GM = 1024 / BM;
    const int tiles_n = (int)gridDim.x / KS / tiles_m;
    const int band = wg / (GM * tiles_n), within = wg % (GM * tiles_n), band_m = min(GM, tiles_m - band * GM);
    const int t0 = (band * GM + within % band_m) * BM, tn = G.tn0 + within / band_m;
    const int set = (tn >= G.tile0[1] ? 1 : 0) + (tn >= G.tile0[2] ? 1 : 0);
    const int n0 = (tn - (set == 0 ? 0 : (set == 1 ? G.tile0[1] : G.tile0[2]))) * BNE;
    const int N = set == 0 ? P.rows[0] : (set == 1 ? P.rows[1] : P.rows[2]);
    const uint8_t *__restrict__ W = set == 0 ? P.W[0] : (set == 1 ? P.W[1] : P.W[2]);
    const half_t *__restrict__ bias = set == 0 ? P.bias[0] : (set == 1 ? P.bias[1] : P.bias[2]);
    const int ksteps = G.K / BK;
    const int s0 = (int)((long long)kz * ksteps / KS), nsteps = (int)((long long)(kz + 1) * ksteps / KS) - s0;      // (parts of K may differ by one step)

    // ---- sources of this wave's direct-to-LDS pieces
    // activations: 4 pieces of 8 token rows x 128 bytes; lane -> (row, chunk), the swizzle applied to the SOURCE (the load writes LDS linearly)
    const half_t *xsrc[XP];
#pragma unroll
    for (int j = 0; j < XP; j++) {
        const int row = (wave * XP + j) * 8 + lane / 8;
        const int c = (lane % 8) ^ swz(row);
        xsrc[j] = X + (size_t)min(t0 + row, T - 1) * P.ldx + c * 8;
    }
    // weights: this wave's 32 rows = two 16-row tiles of the MO copy (ifa_gemm_rows_mfma.h: per tile, superstep S of 128 columns is ONE
    // contiguous KiB -- lane 16 g + r holds the 16 code bytes of block 4 S + g of row r -- so a tile's K range is one linear stream,
    // like a decode row; the tiled rows fetched 32 bytes per row and step measured 1 TB/s: 32 DRAM pages per request).  A step is
    // half a superstep: lane -> (tile l / 32, block (l / 16) % 2 of the half, row l % 16), 2 x 512 contiguous bytes per request.
    const int nsup = nblk >> 2, nq4 = (nsup + 3) >> 2;
    const size_t mo_tile = (size_t)(nsup + nq4) * 1024;
    const uint8_t *wsrc, *wsbs;
    {
        const int nl0 = wave * 32;
        const uint8_t *Wm = (GLU && nl0 >= BNE) ? P.W1 : W;
        const int row0 = n0 + (GLU && nl0 >= BNE ? nl0 - BNE : nl0);
        const int tile16 = min((row0 >> 4) + (lane >> 5), ((N + 15) >> 4) - 1);
        wsrc = Wm + (size_t)tile16 * mo_tile + (size_t)(lane & 31) * 16;
        wsbs = wsrc + (size_t)nsup * 1024;
    }
    // the six requests of a stage, one at a time (piece 0..3: activation rows, 4: codes, 5: (base, scale) words): the main loop places
    // them between its MFMAs
    auto issue_piece = [&](int step, int slot, int piece) {
        char *st = smem + (size_t)slot * MD_STAGE;
        if (piece < XP) {
#if IFA_MID_ABL != 1
            __builtin_amdgcn_global_load_lds((md_glb_t *)(xsrc[piece] + (size_t)(s0 + step) * BK), (md_lds_t *)(st + (size_t)(wave * XP + piece) * 1024), 16, 0, 0);
#endif
            return;
        }
#if IFA_MID_ABL != 5
        const int S = (s0 + step) >> 1, half = (s0 + step) & 1;
        if (piece == XP)
            __builtin_amdgcn_global_load_lds((md_glb_t *)(wsrc + (size_t)S * 1024 + (size_t)half * 512), (md_lds_t *)(st + MD_A_BYTES + wave * 1024), 16, 0, IFA_MID_WAUX);
        else
            __builtin_amdgcn_global_load_lds((md_glb_t *)(wsbs + (size_t)(S >> 2) * 1024 + (size_t)half * 512 + (size_t)(S & 3) * 4),
                                             (md_lds_t *)(st + MD_A_BYTES + MD_WC_BYTES + wave * 256), 4, 0, IFA_MID_WAUX);
#endif
    };
    auto issue = [&](int step, int slot) {
#pragma unroll
        for (int pc = 0; pc < XP + 2; pc++) issue_piece(step, slot, pc);
    };

    // ---- per-lane LDS offsets: lane (i, g) reads token row i of a 32-row tile, chunk 2 ks + g; weight row i of its wave
    const int i = lane & 31, g = lane >> 5;
    const int lc = g ^ swz(i);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(md_lds_t *)smem;
    const uint32_t a_off = lds0 + (uint32_t)(i * ROWB);
    // (the stage holds this wave's codes as [tile][block of the half][row] x 16 bytes, the words likewise x 4 bytes)
    const uint32_t wc_off = lds0 + (uint32_t)(MD_A_BYTES + wave * 1024 + ((i >> 4) * 32 + (i & 15)) * 16);
    const uint32_t ws_off = lds0 + (uint32_t)(MD_A_BYTES + MD_WC_BYTES + wave * 256 + ((i >> 4) * 32 + (i & 15)) * 4);
    const float fp8_up = q4_fp8_up();

    md_f16v acc[TA];
#pragma unroll
    for (int a = 0; a < TA; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.0f;

    constexpr bool LD = IFA_MID_LOADERS != 0;
    if (!LD || loader) {
#pragma unroll
        for (int p = 0; p < NS - 1; p++) issue(min(p, nsteps - 1), p);
    }
    int slot = 0;
    if constexpr (LD) {
        // The loader wave of a SIMD (IFA_MID_LOADERS): the six requests of a stage took their issue slots (~70 cycles each, ablation
        // builds: profiles/r06_prefill_mid_parts.log) out of the ONE instruction stream that also issues the MFMAs; a second wave on
        // the SIMD issues them on the memory port while the computing wave multiplies.  Same ring protocol, same barrier: the loader
        // waits for ITS requests of stage `step` (vmcnt is per wave), every wave meets at the barrier, the loader requests stage
        // step + NS - 1 into the slot the computing waves have just left.
        if (loader) {
            for (int step = 0; step < nsteps; step++) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * MD_VM) : "memory");
                __builtin_amdgcn_s_barrier();
                const int nslot = slot == 0 ? NS - 1 : slot - 1;
                issue(min(step + NS - 1, nsteps - 1), nslot);
                slot = slot + 1 == NS ? 0 : slot + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the repeats behind the last stage have landed before the ring is reused
            __builtin_amdgcn_s_barrier();                          // (pairs with the computing waves' first barrier behind the loop)
            return;
        }
    }
    if (trc) trc[1] = wall_clock64();
    for (int step = 0; step < nsteps; step++) {
        // stage `step` has landed (the NS - 2 younger stages may still be in flight); the barrier makes every wave's pieces visible
        // and says that every wave is done with the stage read last step, whose slot takes the request of stage step + NS - 1
        // (the bare barrier: __syncthreads() is fence + barrier, and the fence waits vmcnt(0) -- for the stages just requested)
        if constexpr (!LD) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * MD_VM) : "memory");
#if IFA_MID_ABL != 4
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        const int nslot = slot == 0 ? NS - 1 : slot - 1, nstep = min(step + NS - 1, nsteps - 1);      // the stage requested during this step
        const uint32_t sb = (uint32_t)(slot * MD_STAGE);
        // A wave alone on its SIMD issues in order: request issue (6 x ~80 cycles), the 19 LDS reads and their latency, the conversion and
        // the MFMAs ran one after the other (0.65 us per step, IFA_MID_TRACE; the MFMAs alone are 0.21).  Now only the raw weights and the
        // first group's activation fragments are read in front of the MFMAs; the other groups' fragments, the next stage's requests and
        // the next group's conversion go BETWEEN the MFMAs (sched_barrier pins the order).
        md_u4 cw[2]; md_u2 sw; md_u4 fa[4][TA];
        cw[0] = md_lds_b128(wc_off + sb);
        cw[1] = md_lds_b128(wc_off + sb + 256);
        sw[0] = md_lds_b32(ws_off + sb);
        sw[1] = md_lds_b32(ws_off + sb + 64);
#pragma unroll
        for (int a = 0; a < TA; a++) fa[0][a] = md_lds_b128(a_off + sb + (uint32_t)(a * 32 * ROWB) + (uint32_t)(((0) ^ lc) << 4));
        // fragments are requested LEAD 16-column groups ahead (round 6, second pass): with one group of lead and TA = 4 the wave sat in
        // lgkmcnt(0) at the end of every group -- four waves read 64 KB of LDS per step, 512 cycles of the LDS pipe, as long as the step's
        // MFMAs -- so the waits are counted (LDS returns in order; nothing else of this loop counts on lgkmcnt; the counter holds 15) and
        // the pipe never drains.  TA = 8: a group's eight MFMAs cover the next group's reads, one group of lead.
        constexpr int LEAD = (IFA_MID_PF >= 2 && TA <= 4) ? 2 : 1;
        if constexpr (LEAD == 2) {
#pragma unroll
            for (int a = 0; a < TA; a++) fa[1][a] = md_lds_b128(a_off + sb + (uint32_t)(a * 32 * ROWB) + (uint32_t)(((2) ^ lc) << 4));
        }
        // (a counted wait names the registers it makes valid: later uses depend on the statement)
        auto landed = [&](int k) {
#pragma unroll
            for (int a = 0; a < TA; a++) asm volatile("" : "+v"(fa[k][a]));
        };
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(cw[0]), "+v"(cw[1]), "+v"(sw) : "n"(TA * LEAD));
        // this lane's 8 weights of group ks: chunk 2 (ks % 2) + g of block ks / 2 -- value half(fma(q, scale, base)), q4x8_dequant's
        // arithmetic (ifa_dequant_q4.h) in four pieces
        uint32_t d_lo = 0, d_hi = 0; q4_f2 d_s2 = {0, 0}, d_b2 = {0, 0}, d_e02 = {0, 0}, d_e46 = {0, 0}, d_o13 = {0, 0}, d_o57 = {0, 0};
        auto dq_a = [&](int ks) {
            const int h = ks >> 1;
            const uint32_t sbw = sw[h];
            const float base = hbits2f((uint16_t)(sbw & 0xFFFFu)), scale = hbits2f((uint16_t)(sbw >> 16)) * fp8_up;
            const uint32_t c_lo = cw[h][2 * (ks & 1)], c_hi = cw[h][2 * (ks & 1) + 1];      // (scalars first: one select, not an indexed element read)
            const uint32_t code = g ? c_hi : c_lo;
            d_lo = code & 0x0F0F0F0Fu; d_hi = (code >> 4) & 0x0F0F0F0Fu;
            d_s2 = q4_f2{scale, scale}; d_b2 = q4_f2{base, base};
        };
#ifndef IFA_MID_SCALAR_FMA
#define IFA_MID_SCALAR_FMA 0      // 1: the conversion's fused multiply-adds as scalar v_fma_f32 (the guide prices a packed f32 op beside MFMAs at +22 cycles)
#endif
        auto fma2 = [&](q4_f2 x) {
#if IFA_MID_SCALAR_FMA
            q4_f2 r;
            float r0, r1;
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(x[0]), "v"(d_s2[0]), "v"(d_b2[0]));
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(x[1]), "v"(d_s2[0]), "v"(d_b2[0]));
            r[0] = r0; r[1] = r1;
            return r;
#else
            return __builtin_elementwise_fma(x, d_s2, d_b2);
#endif
        };
        auto dq_b = [&]() {
            d_e02 = fma2(__builtin_amdgcn_cvt_pk_f32_fp8((int)d_lo, false));
            d_e46 = fma2(__builtin_amdgcn_cvt_pk_f32_fp8((int)d_lo, true));
        };
        auto dq_c = [&]() {
            d_o13 = fma2(__builtin_amdgcn_cvt_pk_f32_fp8((int)d_hi, false));
            d_o57 = fma2(__builtin_amdgcn_cvt_pk_f32_fp8((int)d_hi, true));
        };
        auto dq_d = [&]() {
            const q4_h2 w0 = __builtin_convertvector(q4_f2{d_e02[0], d_o13[0]}, q4_h2), w1 = __builtin_convertvector(q4_f2{d_e02[1], d_o13[1]}, q4_h2);
            const q4_h2 w2 = __builtin_convertvector(q4_f2{d_e46[0], d_o57[0]}, q4_h2), w3 = __builtin_convertvector(q4_f2{d_e46[1], d_o57[1]}, q4_h2);
            return md_h8{w0[0], w0[1], w1[0], w1[1], w2[0], w2[1], w3[0], w3[1]};
        };
        auto frag = [&](int ks, int a) { fa[ks][a] = md_lds_b128(a_off + sb + (uint32_t)(a * 32 * ROWB) + (uint32_t)(((2 * ks) ^ lc) << 4)); };
        md_h8 fbc, fbn;
        dq_a(0); dq_b(); dq_c(); fbc = dq_d();
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(TA * (LEAD - 1)));
        landed(0);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#if IFA_MID_ABL == 2
#pragma unroll
            for (int a = 0; a < TA; a++) acc[a][ks] += (float)fbc[a % 8] * (float)__builtin_bit_cast(md_h8, fa[ks][a])[0];
            if (ks < 3) { dq_a(ks + 1); dq_b(); dq_c(); fbn = dq_d(); }
            if (ks + LEAD < 4) for (int a = 0; a < TA; a++) frag(ks + LEAD, a);
            if (ks < 3 && !LD) for (int pc = ks * TA; pc < (ks + 1) * TA && pc < XP + 2; pc++) issue_piece(nstep, nslot, pc);
#else
            // TA MFMAs; between them (sched_barrier pins the order): the fragment reads of group ks + LEAD two at a time, the four pieces of
            // the next group's conversion, and -- without loader waves -- the next stage's requests
#pragma unroll
            for (int a = 0; a < TA; a++) {
                __builtin_amdgcn_sched_barrier(0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(md_h8, fa[ks][a]), fbc, acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + LEAD < 4 && 2 * a + 1 < TA) { frag(ks + LEAD, 2 * a); frag(ks + LEAD, 2 * a + 1); }
                if (ks < 3) {
                    if (a == 0) dq_a(ks + 1);
                    else if (a == TA / 4) dq_b();
                    else if (a == TA / 2) dq_c();
                    else if (a == 3 * TA / 4) fbn = dq_d();
                    if constexpr (!LD) { if (ks * TA + a < XP + 2) issue_piece(nstep, nslot, ks * TA + a); }
                }
            }
#endif
            if (ks < 3) {
                // group ks + 1 has landed; with two groups of lead the TA requests of group ks + 2 may still be in flight
                if (LEAD == 2 && ks + 2 < 4) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(TA));
                else asm volatile("s_waitcnt lgkmcnt(0)");
                landed(ks + 1);
            }
            fbc = fbn;
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    // the repeats requested behind the last stage must have landed before the ring's memory is used for anything else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (LD) __builtin_amdgcn_s_barrier();      // (the loader waves' last barrier: their requests have landed; they end there)
    if (trc) trc[2] = wall_clock64();
    // ---- parts of K as a REDUCE-SCATTER (KS > 1).  A thread's 64 sums are 16 units of 16 bytes; unit u = 4 a + q holds the 8 token
    // rows [8 u, 8 u + 8) of this lane's weight row.  Part kz OWNS units [kz UB, (kz + 1) UB), i.e. token rows [8 kz UB, 8 (kz + 1) UB) of the
    // tile: it stores its other units (write-through, 16 bytes per store), arrives at the tile's counter, waits for all KS
    // parts, adds the other parts' values of ITS units in K order (part 0 + part 1 + ... with its own at position kz: the same
    // sequence of additions whichever part performs it) and runs the epilogue for its rows.  k_gemm_big's form -- the last part
    // reads every other part's whole tile -- measured 8 us per launch here (192 KB through one CU) and limits KS; this one reads
    // (KS - 1) / KS x 64 KB per workgroup whatever KS is, so short products can be cut into enough parts to fill the chip twice.
    constexpr int NU = 4 * TA;                           // 8-token-row bands of the tile = 16-byte units of a thread's sums
    static_assert(NU % KS == 0, "parts of K must divide the tile's row bands");
    constexpr int UB = NU / KS;                          // units (8-row bands) a part owns
    const int row_lo = KS == 1 ? 0 : kz * UB * 8, nrows_own = KS == 1 ? BM : UB * 8;
    constexpr int VEC = BNE / 8;
    constexpr int PIECES = (BM / KS * VEC + NT - 1) / NT;  // 16-byte output pieces per thread (its part's rows)
    // GM_RESIDUAL: this thread's pieces of the residual rows are requested NOW: issued from the output loop they were dependent round
    // trips at the very end of the launch (4.6 us of a 25 us launch, IFA_MID_TRACE)
    md_h8 resv[EPI == GM_RESIDUAL ? PIECES : 1];
    if constexpr (EPI == GM_RESIDUAL) {
        const int vr0 = (set >= 1 ? P.rows[0] : 0) + (set >= 2 ? P.rows[1] : 0);
#pragma unroll
        for (int p = 0; p < PIECES; p++) {
            const int idx = tid + p * NT, tl = row_lo + min(idx / VEC, nrows_own - 1), v = idx % VEC;
            const int tok = min(t0 + tl, T - 1), n = min(n0 + v * 8, N - 8);
            resv[p] = *reinterpret_cast<const md_h8 *>(P.res + (size_t)tok * P.ldres + (P.Yset[0] ? 0 : vr0) + n);
        }
    }
    if constexpr (KS > 1) {
        char *pt = reinterpret_cast<char *>(G.part) + (size_t)wg * ((size_t)KS * NU * NT * 16);
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(pt, 0, KS * NU * NT * 16, 0x00020000);
        unsigned *arrive = G.flags + 2 * wg, *done = arrive + 1;
#pragma unroll
        for (int u = 0; u < NU; u++) {
            if (u / UB == kz) continue;                  // (wave-uniform)
            md_u4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) { const float f = acc[u >> 2][4 * (u & 3) + e]; v[e] = __builtin_bit_cast(uint32_t, f); }
            __builtin_amdgcn_raw_buffer_store_b128(v, prs, ((kz * NU + u) * NT + tid) * 16, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t_wait = wall_clock64();
            while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)KS) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t_wait > WAIT_TIMEOUT_TICKS) {      // the other parts are not resident: leave a code and go on (the host fails the call)
                    if (G.err) __hip_atomic_store(G.err, 0x82u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; u++) {
            if (u / UB != kz) continue;
            md_u4 pv[KS];
#pragma unroll
            for (int z = 0; z < KS; z++) pv[z] = md_u4{0, 0, 0, 0};
#pragma unroll
            for (int z = 0; z < KS; z++)
                if (z != kz) pv[z] = __builtin_amdgcn_raw_buffer_load_b128(prs, ((z * NU + u) * NT + tid) * 16, 0, 16);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float own = acc[u >> 2][4 * (u & 3) + e];
                float sf = 0.0f;
#pragma unroll
                for (int z = 0; z < KS; z++) {
                    const uint32_t uz = pv[z][e];            // (scalars first: a bit_cast of a vector ELEMENT reads element 0 every time)
                    const float term = z == kz ? own : __builtin_bit_cast(float, uz);
                    sf = z == 0 ? term : sf + term;
                }
                acc[u >> 2][4 * (u & 3) + e] = sf;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the part that is last to finish reading puts both counters back to zero (the next launch on this stream finds them clean)
        if (tid == 0) {
            const unsigned d = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == (unsigned)(KS - 1)) {
                __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    // ---- epilogue (k_gemm_big's arithmetic, this part's rows): products -> F16 (+ bias) -> an LDS tile (rows padded by 64 bytes) ->
    // 16-byte pieces of output rows, residual / GLU applied on the way out
    if (trc) trc[3] = wall_clock64();
    half_t *__restrict__ Yo; int ldo, vrow0 = 0;
    if (P.Yset[0]) { Yo = set == 0 ? P.Yset[0] : (set == 1 ? P.Yset[1] : P.Yset[2]); ldo = set == 0 ? P.ldyset[0] : (set == 1 ? P.ldyset[1] : P.ldyset[2]); }
    else { Yo = P.Y; ldo = P.ldy; vrow0 = (set >= 1 ? P.rows[0] : 0) + (set >= 2 ? P.rows[1] : 0); }
    constexpr int CROW = BN * 2 + 64;
    __syncthreads();                                    // every wave is done with the ring
    {
        const bool odd = i & 1;
        const int cl = wave * 32 + i;
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
                if (KS > 1 && (4 * a + (rp >> 1)) / UB != kz) continue;      // (this part's units only; wave-uniform)
                half_t y0 = f2h(acc[a][2 * rp]), y1 = f2h(acc[a][2 * rp + 1]);
                if (hb) { y0 = f2h(h2f(y0) + bv); y1 = f2h(h2f(y1) + bv); }
                const uint32_t u0 = __builtin_bit_cast(uint16_t, y0), u1 = __builtin_bit_cast(uint16_t, y1);
                const uint32_t send = odd ? u0 : u1;
                const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, false);   // quad_perm [1, 0, 3, 2]
                const uint32_t packed = odd ? (recv | (u1 << 16)) : (u0 | (recv << 16));
                const int rr = 2 * rp + (odd ? 1 : 0);
                const int tl = a * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * g;
                *reinterpret_cast<uint32_t *>(smem + (size_t)tl * CROW + (size_t)(cl & ~1) * 2) = packed;
            }
    }
    __syncthreads();
    {
#pragma unroll
        for (int p = 0; p < PIECES; p++) {
            const int idx = tid + p * NT, tr = idx / VEC, v = idx % VEC, tl = row_lo + tr, tok = t0 + tl, n = n0 + v * 8;
            if (tr >= nrows_own) continue;
            md_h8 y = *reinterpret_cast<const md_h8 *>(smem + (size_t)tl * CROW + (size_t)v * 16);
            if constexpr (GLU) {
                const md_h8 u = *reinterpret_cast<const md_h8 *>(smem + (size_t)tl * CROW + (size_t)(BNE + v * 8) * 2);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const half_t act = f2h(act_fn(h2f(y[e]), P.act_kind));
                    y[e] = f2h(h2f(act) * h2f(u[e]));
                }
            } else if constexpr (EPI == GM_RESIDUAL) {
#pragma unroll
                for (int e = 0; e < 8; e++) y[e] = f2h(h2f(resv[p][e]) + h2f(y[e]));
            }
            if (tok < T && n < N) *reinterpret_cast<md_h8 *>(Yo + (size_t)tok * ldo + vrow0 + n) = y;
        }
    }
    if (trc) trc[4] = wall_clock64();
}

static int md_num_cus()
{
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t prop; n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
    return n;
}

bool gemm_mid_ok(int w_dtype, const GmArgs &P, int epi)
{
    (void)w_dtype;      // (any format the MO copy holds: 4-bit codes with value q * scale + base per 32 weights)
    if (!P.mo || P.T < 1 || P.nsets < 1 || P.nsets > 3 || P.nblk < 8 || P.nblk % 4 != 0) return false;
    for (int i = 0; i < P.nsets; i++) if (P.rows[i] % 16 != 0) return false;
    if (epi == GM_GLU && (P.nsets != 1 || !P.W1 || P.rows[0] % 8 != 0)) return false;
    for (int i = 0; i < P.nsets; i++) if (P.rows[i] % 8 != 0) return false;
    if ((P.Yset[0] ? (P.ldyset[0] | P.ldyset[1] | P.ldyset[2]) : P.ldy) % 8 != 0 || (epi == GM_RESIDUAL && P.ldres % 8 != 0) || P.ldx % 8 != 0) return false;
    return true;
}

template <int EPI, int KS, int NS, int TA>
static int md_run(const GmArgs &P, BigGeo G, int tn_count, hipStream_t s)
{
    if constexpr (KS > 1) {
        G.err = wait_err_word();
        const size_t tiles = (size_t)G.tiles_m * tn_count, part_bytes = tiles * (size_t)KS * (32 * TA) * MD_BN * 4;
        void *scratch = nullptr;
        int rcs = gemm_splitk_scratch(s, part_bytes, 2 * tiles, &scratch);      // (two counters per tile)
        if (rcs) return rcs;
        G.flags = (unsigned *)scratch; G.part = (unsigned long long *)((char *)scratch + SPLITK_FLAG_BYTES_H);
    }
    const size_t smem = std::max((size_t)NS * md_stage(TA), (size_t)(32 * TA) * (MD_BN * 2 + 64));
    auto kern = k_gemm_mid<EPI, KS, NS, TA>;
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    const long long grid = (long long)G.tiles_m * tn_count * KS;
    if constexpr (KS > 1) {
        if (!wait_grid_fits((const void *)kern, MD_THREADS, smem, grid)) return 1;      // (1: not launched -- the caller takes fewer parts)
    }
    static const bool tracing = getenv("IFA_MID_TRACE") != nullptr;
    if (tracing) {      // measurement: per-workgroup phase stamps of this launch on stderr (synchronous)
        static long long *buf = nullptr;
        if (!buf) (void)hipMalloc((void **)&buf, sizeof(long long) * 8 * 4096);
        if (buf && grid <= 4096) {
            (void)hipMemsetAsync(buf, 0, sizeof(long long) * 8 * (size_t)grid, s);
            GmArgs Q = P; Q.trace = buf;
            kern<<<dim3((unsigned)grid), dim3(MD_THREADS), smem, s>>>(Q, G);
            std::vector<long long> h((size_t)grid * 8);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
            long long t0 = h[0];
            for (long long b = 0; b < grid; b++) t0 = std::min(t0, h[(size_t)b * 8]);
            double mx[5] = {0, 0, 0, 0, 0}, sm[5] = {0, 0, 0, 0, 0};
            for (long long b = 0; b < grid; b++) for (int k = 0; k < 5; k++) { const double v = (h[(size_t)b * 8 + k] - t0) * 0.01; mx[k] = std::max(mx[k], v); sm[k] += v; }
            fprintf(stderr, "k_gemm_mid<%d,%d,%d,%d> grid %lld steps %d | us after the first workgroup's start, mean (max): start %.2f (%.2f) loop begins %.2f (%.2f) loop ends %.2f (%.2f) parts summed %.2f (%.2f) end %.2f (%.2f)\n",
                    EPI, KS, NS, TA, grid, G.K / MD_BK / KS, sm[0] / grid, mx[0], sm[1] / grid, mx[1], sm[2] / grid, mx[2], sm[3] / grid, mx[3], sm[4] / grid, mx[4]);
            return IFA_OK;
        }
    }
    kern<<<dim3((unsigned)grid), dim3(MD_THREADS), smem, s>>>(P, G);
    return IFA_OK;
}

#ifndef IFA_MID_NS
#define IFA_MID_NS 3
#endif

template <int EPI, int TA>
static int md_launch(const GmArgs &P, hipStream_t s)
{
    constexpr int MD_BM = 32 * TA;
    constexpr int BNE = EPI == GM_GLU ? MD_BN / 2 : MD_BN;
    BigGeo G; memset(&G, 0, sizeof(G));
    G.tile0[0] = 0;
    for (int i = 0; i < 3; i++) G.tile0[i + 1] = G.tile0[i] + (i < P.nsets ? (P.rows[i] + BNE - 1) / BNE : 0);
    const int tn_count = G.tile0[P.nsets];
    for (int i = P.nsets; i < 3; i++) G.tile0[i] = 1 << 30;
    G.tiles_m = (P.T + MD_BM - 1) / MD_BM; G.K = P.nblk * 32; G.tn0 = 0;
    const int ksteps = G.K / MD_BK;
    // parts of K: the most that still put ONE workgroup on a CU (two fit -- LDS, registers -- but a grid between one and two per CU
    // ends when the doubly loaded CUs do: wq | wk | wv of 128 tokens, 96 tiles: 384 workgroups 32.3 us, 192 workgroups 27; of 256
    // tokens, 192 tiles: one part beats two; wo / w2 of one token tile, 32 tiles: eight parts, 256 workgroups, beat four -- 128 tokens
    // 5.34 -> 4.86 ms in all, profiles/r06_prefill_mid_parts.log), every part >= MINS steps
    static const bool two_per_cu = getenv("IFA_MID_TWO_PER_CU") != nullptr;      // (measurement: the first form's limit)
    // (the gated pair keeps two per CU: 172 tiles of one token tile as 344 workgroups 45.7 us, as 172 workgroups 50.5)
    const long long tiles = (long long)G.tiles_m * tn_count, cap = ((two_per_cu || EPI == GM_GLU) ? 2ll : 1ll) * md_num_cus();
    static const bool no_glu_split = getenv("IFA_MID_NO_GLU_SPLIT") != nullptr;      // (measurement: the gated pair as ONE part of K, like k_gemm_big's -- bit-identical products)
    const bool may = !P.no_waits && waits_enabled() && !(EPI == GM_GLU && no_glu_split);
    int rc = 1;
    static const int force_all = getenv("IFA_MID_KS") ? atoi(getenv("IFA_MID_KS")) : 0;      // (measurement: at most this many parts)
    static const int force_epi = getenv(EPI == GM_PLAIN ? "IFA_MID_KS_PLAIN" : (EPI == GM_GLU ? "IFA_MID_KS_GLU" : "IFA_MID_KS_RES"))
                                     ? atoi(getenv(EPI == GM_PLAIN ? "IFA_MID_KS_PLAIN" : (EPI == GM_GLU ? "IFA_MID_KS_GLU" : "IFA_MID_KS_RES"))) : 0;
    const int force_ks = force_epi ? force_epi : force_all;
    static const int mins = getenv("IFA_MID_MINSTEPS") ? atoi(getenv("IFA_MID_MINSTEPS")) : 4;
    auto fits = [&](int ks) { return may && tiles * ks <= cap && ksteps / ks >= mins && (!force_ks || ks <= force_ks); };
    // (16 parts: the exchanged sums grow with the part count -- 30 MB written and read for wo: 30 us against 17 at 8 -- a measurement setting)
    if constexpr (TA == 4) { if (force_ks >= 16 && fits(16)) rc = md_run<EPI, 16, IFA_MID_NS, TA>(P, G, tn_count, s); }
    if (rc == 1 && fits(8)) rc = md_run<EPI, 8, IFA_MID_NS, TA>(P, G, tn_count, s);
    if (rc == 1 && fits(4)) rc = md_run<EPI, 4, IFA_MID_NS, TA>(P, G, tn_count, s);
    if (rc == 1 && fits(2)) rc = md_run<EPI, 2, IFA_MID_NS, TA>(P, G, tn_count, s);
    if (rc == 1) rc = md_run<EPI, 1, IFA_MID_NS, TA>(P, G, tn_count, s);
    if (rc) return rc;
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int gemm_mid(const GmArgs &P, int epi, hipStream_t s)
{
    if (!gemm_mid_ok(Q4_B32T1A, P, epi)) return ifa_fail(IFA_ERR_ARG, "mid-length GEMM: %d blocks per row / epilogue %d", P.nblk, epi);
    // 256-token tiles (TA = 8): the weight stream and its conversion serve twice the products per step
    static const int force_ta = getenv("IFA_MID_TA") ? atoi(getenv("IFA_MID_TA")) : 0;      // (measurement)
    // (measured round 6: 512 / 768 tokens 10.96 / 16.48 ms with 128-token tiles, 11.37 / 17.50 with 256; 1024 tokens 19.57 / 19.25 -- the
    //  wide tile pays only where the large-tile kernel already ties, so it stays a measurement setting)
    const bool wide = force_ta == 8;
    if (wide) {
        if (epi == GM_PLAIN) return md_launch<GM_PLAIN, 8>(P, s);
        if (epi == GM_RESIDUAL) return md_launch<GM_RESIDUAL, 8>(P, s);
        return md_launch<GM_GLU, 8>(P, s);
    }
    if (epi == GM_PLAIN) return md_launch<GM_PLAIN, 4>(P, s);
    if (epi == GM_RESIDUAL) return md_launch<GM_RESIDUAL, 4>(P, s);
    return md_launch<GM_GLU, 4>(P, s);
}

} // namespace ifa
