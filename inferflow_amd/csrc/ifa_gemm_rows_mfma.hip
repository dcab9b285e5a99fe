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
#include "ifa_gemm_rows_mfma_body.h"

namespace ifa {

// MO layout of one matrix from its tiled copy: thread = (tile, superstep S, lane (r, g)) copies the 16 code bytes of the 32 weights
// 32 (4S + g) .. + 31 of row 16 tile + r and their (base, scale) word; rows past the end: zero codes and words (weight 0).
// B64: the source is a nibble-pair format with ONE (base, scale) per 64 weights -- Q4_B64T1, and Q3H_B64T1 as it is streamed
// (ifa_tiled.h: [32 B codes] x n, [base, scale] x n per row) -- whose value is q * scale + base like Q4_B32T1: the 64-block's
// word is written for both of its halves, and the rows GEMM runs unchanged on 32-weight blocks.
template <bool B64>
__global__ void __launch_bounds__(256) k_gemm_rows_mo_build(const uint8_t *__restrict__ tiled, int rows, int nblk, size_t row_bytes, uint8_t *__restrict__ mo)
{
    const int nsup = nblk >> 2, nq4 = (nsup + 3) >> 2;            // nblk: 32-weight blocks per row
    const size_t mo_tile = (size_t)(nsup + nq4) * 1024;
    const int ntiles = (rows + 15) >> 4;
    const size_t total = (size_t)ntiles * nq4 * 4 * 64;            // (supersteps padded to whole quads: the pad's words are zero)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t ts = i >> 6;
        const int S = (int)(ts % (size_t)(nq4 * 4)), tile = (int)(ts / (size_t)(nq4 * 4));
        const int r = lane & 15, g = lane >> 4;
        const int row = tile * 16 + r, blk = 4 * S + g;
        const bool in = row < rows && S < nsup;
        u32x4 c = {0, 0, 0, 0};
        uint32_t w = 0;
        if (in) {
            const uint8_t *rp = tiled + (size_t)row * row_bytes;
            if constexpr (B64) {
                const int n64 = nblk >> 1;
                c = *reinterpret_cast<const u32x4 *>(rp + (size_t)(blk >> 1) * 32 + (size_t)(blk & 1) * 16);
                w = *reinterpret_cast<const uint32_t *>(rp + (size_t)n64 * 32 + (size_t)(blk >> 1) * 4);
            } else {
                c = *reinterpret_cast<const u32x4 *>(rp + (size_t)blk * 16);
                w = *reinterpret_cast<const uint32_t *>(rp + (size_t)nblk * 16 + (size_t)blk * 4);
            }
        }
        uint8_t *tb = mo + (size_t)tile * mo_tile;
        if (S < nsup) *reinterpret_cast<u32x4 *>(tb + (size_t)S * 1024 + (size_t)lane * 16) = c;
        *reinterpret_cast<uint32_t *>(tb + (size_t)(nsup + (S >> 2)) * 1024 + (size_t)lane * 16 + (size_t)(S & 3) * 4) = w;
    }
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
    P.mo = 0; P.trace = nullptr; P.kparts = 0; P.kpart_sums = nullptr;
    gemm_rows_mfma_body<MAXT, TX, GM_PLAIN, 0, false, 0>(P, smem);
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
    auto kern = k_gemm_rows_mfma<MT, TX, EPI, NORM, false, 0>;
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
    if (P.T < 2 || P.T > (P.mo ? 32 : 16) || P.nblk <= 0 || P.nblk % 4 != 0 || P.nsets < 1 || P.nsets > 3) return false;      // (17..32 rows: MO layout only)
    if (norm == 1 && (P.nblk * 32 > GmGeo<32>::CHUNK_COLS || P.T > (P.mo ? 16 : 8))) return false;      // the norm prologue: whole rows in one chunk (tiled layout: <= 8 rows)
    if (epi == GM_GLU && (P.nsets != 1 || !P.W1)) return false;
    for (int i = 0; i < P.nsets; i++) if (P.rows[i] <= 0 || (P.nsets > 1 && P.rows[i] % 16 != 0)) return false;
    for (int i = 0; i < P.nsets; i++)      // 32-bit byte offsets inside a matrix (tiled and MO copies are 20 bytes per block + padding)
        if (((size_t)P.rows[i] + 16) * ((size_t)P.nblk + 16) * 20 >= ((size_t)1 << 32)) return false;
    return true;
}

int gemm_rows_mfma_launch(const GmArgs &P0, int epi, int norm, hipStream_t s)
{
    GmArgs P = P0;
    P.total_rows = 0;
    for (int i = 0; i < P.nsets; i++) P.total_rows += P.rows[i];
    if (!gemm_rows_mfma_fused_ok(P, epi, norm)) return ifa_fail(IFA_ERR_ARG, "rows GEMM: shape not covered (T %d, %d blocks, %d sets)", P.T, P.nblk, P.nsets);
    int wgs, maxt;
    static const int per_cu = getenv("IFA_ROWS_WGS_PER_CU") ? std::max(1, atoi(getenv("IFA_ROWS_WGS_PER_CU"))) : 1;     // tuning aid
    gm_geometry((size_t)P.total_rows, gm_num_cus() * (P.T <= 8 ? per_cu : 1), &wgs, &maxt);
    if (epi == GM_GLU) {                       // pairs of tiles: 1, 2, 3, 4 pairs per workgroup
        if (maxt > 4) { maxt = 4; wgs = ((P.total_rows + 15) / 16 + 3) / 4; }
        maxt = maxt == 3 ? 6 : maxt * 2;
    }
    const size_t smem = gm_smem(P.T, maxt, P.mo);
    static const bool tracing = getenv("IFA_ROWS_TRACE") != nullptr;     // tuning aid: eager launches only (it synchronises)
    static long long *trace_dev = nullptr;
    if (tracing) {
        if (!trace_dev) IFA_HIP_CHECK(hipMalloc(&trace_dev, 4096 * 32 * sizeof(long long)));
        IFA_HIP_CHECK(hipMemsetAsync(trace_dev, 0, 4096 * 32 * sizeof(long long), s));
        P.trace = trace_dev;
    }
    int rc;
    if (P.mo) rc = gemm_rows_mo_launch(P, epi, norm, wgs, maxt, s);
    else switch (maxt) {
    case 1: rc = gm_launch1<1>(wgs, smem, P, epi, norm, s); break;
    case 2: rc = gm_launch1<2>(wgs, smem, P, epi, norm, s); break;
    case 3: rc = gm_launch1<3>(wgs, smem, P, epi, norm, s); break;
    case 4: rc = gm_launch1<4>(wgs, smem, P, epi, norm, s); break;
    case 6: rc = gm_launch1<6>(wgs, smem, P, epi, norm, s); break;
    default: rc = gm_launch1<8>(wgs, smem, P, epi, norm, s); break;
    }
    if (tracing && rc == IFA_OK) {
        std::vector<long long> h((size_t)wgs * 32);
        IFA_HIP_CHECK(hipStreamSynchronize(s));
        IFA_HIP_CHECK(hipMemcpy(h.data(), trace_dev, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        long long t0 = h[0];
        for (int w = 0; w < wgs; w++) t0 = std::min(t0, h[(size_t)w * 32]);
        auto med = [&](int k, bool mx) {
            std::vector<double> v;
            for (int w = 0; w < wgs; w++) if (h[(size_t)w * 32 + k]) v.push_back((double)(h[(size_t)w * 32 + k] - t0) * 0.01);
            if (v.empty()) return -1.0;
            std::sort(v.begin(), v.end());
            return mx ? v.back() : v[v.size() / 2];
        };
        double wmin = 1e9, wmax = 0;
        for (int k = 8; k < 16; k++) { wmin = std::min(wmin, med(k, false)); wmax = std::max(wmax, med(k, true)); }
        fprintf(stderr, "rows-trace T=%d rows=%d nblk=%d epi=%d norm=%d mo=%d wgs=%d maxt=%d | start med %.2f max %.2f | x requested %.2f | x staged %.2f | "
                "first group done %.2f | loop end (waves) med-min %.2f max %.2f | partials barrier %.2f (max %.2f) | end %.2f (max %.2f)\n",
                P.T, P.total_rows, P.nblk, epi, norm, P.mo, wgs, maxt, med(0, false), med(0, true), med(1, false), med(2, false), med(3, false), wmin, wmax,
                med(4, false), med(4, true), med(5, false), med(5, true));
        fprintf(stderr, "rows-trace   staging: row sums written %.2f, barrier passed %.2f, rows scaled %.2f, stored %.2f, other groups requested %.2f\n", med(6, false), med(7, false),
                med(16, false), med(17, false), med(18, false));
        fprintf(stderr, "rows-trace   loop end by wave (median over workgroups): %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n", med(8, false), med(9, false), med(10, false),
                med(11, false), med(12, false), med(13, false), med(14, false), med(15, false));
    }
    return rc;
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
    static const bool mo_test = getenv("IFA_ROWS_MO_TEST") != nullptr;      // tuning aid (tools/bench_rows.py): a cached MO copy per weight pointer
    if (mo_test) {
        static std::map<std::pair<const void *, size_t>, void *> cache;      // (never freed: a tuning aid)
        void *&mo = cache[std::make_pair(Wt_tiled, rows * 65537 + cols)];
        if (!mo) {
            IFA_HIP_CHECK(hipMalloc(&mo, gemm_rows_mo_bytes(rows, cols)));
            int rc = gemm_rows_mo_build(Q4_B32T1A, Wt_tiled, rows, cols, mo, s);
            if (rc) return rc;
        }
        P.W[0] = (const uint8_t *)mo; P.mo = 1;
    }
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
    const size_t smem = gm_smem(8, maxt, 0);
    switch (maxt) {
    case 1: return gm_launch_grouped<1>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    case 2: return gm_launch_grouped<2>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    case 3: return gm_launch_grouped<3>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    case 4: return gm_launch_grouped<4>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    case 6: return gm_launch_grouped<6>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    default: return gm_launch_grouped<8>(wgs, max_groups, smem, grp, rows, cols, X, Y, s);
    }
}

// The 64-weight nibble formats (Q4_B64T1; Q3H_B64T1 as streamed) rewritten as Q4_B32T1A reference-layout blocks -- {base, scale,
// 16 code bytes} per 32 weights, the 64-block's word in both halves: the same values q * scale + base -- so that long prompts
// take the large-tile GEMM (ifa_gemm.hip, k_gemm_big: blocks of <= 32 values) instead of the split-K kernel.  Compute-bound
// work: the extra 4 bytes per 64 weights do not matter there.
__global__ void __launch_bounds__(256) k_expand_b64_to_q4b32(const uint8_t *__restrict__ tiled, size_t rows, int n64, size_t row_bytes, uint8_t *__restrict__ out)
{
    const size_t total = rows * (size_t)n64 * 2;                  // 32-weight blocks
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / ((size_t)n64 * 2);
        const int b32 = (int)(i % ((size_t)n64 * 2));
        const uint8_t *rp = tiled + row * row_bytes;
        const u32x4 c = *reinterpret_cast<const u32x4 *>(rp + (size_t)(b32 >> 1) * 32 + (size_t)(b32 & 1) * 16);
        const uint32_t w = *reinterpret_cast<const uint32_t *>(rp + (size_t)n64 * 32 + (size_t)(b32 >> 1) * 4);
        uint32_t *o = reinterpret_cast<uint32_t *>(out + i * 20);
        o[0] = w; o[1] = c[0]; o[2] = c[1]; o[3] = c[2]; o[4] = c[3];
    }
}

int expand_b64_to_q4b32(int dtype, const void *tiled, size_t rows, size_t cols, void *out_aos, hipStream_t s)
{
    IFA_REQUIRE(tiled && out_aos && rows > 0 && cols % 64 == 0 && (dtype == Q4_B64T1 || dtype == Q3H_B64T1), "expand_b64_to_q4b32: format %d, %zu x %zu", dtype, rows, cols);
    const size_t total = rows * (cols / 32);
    k_expand_b64_to_q4b32<<<dim3((unsigned)std::min<size_t>(65535, (total + 255) / 256)), dim3(256), 0, s>>>((const uint8_t *)tiled, rows, (int)(cols / 64),
                                                                                                          tiled_row_bytes(dtype, cols / 64), (uint8_t *)out_aos);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

size_t gemm_rows_mo_bytes(size_t rows, size_t cols)
{
    const size_t nsup = cols / 128, nq4 = (nsup + 3) / 4;
    return (rows + 15) / 16 * (nsup + nq4) * 1024;
}

int gemm_rows_mo_build(int dtype, const void *tiled, size_t rows, size_t cols, void *mo, hipStream_t s)
{
    IFA_REQUIRE(tiled && mo && rows > 0 && cols % 128 == 0, "gemm_rows_mo_build: %zu x %zu", rows, cols);
    const size_t items = (rows + 15) / 16 * ((cols / 128 + 3) / 4 * 4) * 64;
    const dim3 grid((unsigned)std::min<size_t>(65535, (items + 255) / 256)), block(256);
    if (dtype == Q4_B32T1A || dtype == Q4_B32T1B)
        k_gemm_rows_mo_build<false><<<grid, block, 0, s>>>((const uint8_t *)tiled, (int)rows, (int)(cols / 32), tiled_row_bytes(dtype, cols / 32), (uint8_t *)mo);
    else if (dtype == Q4_B64T1 || dtype == Q3H_B64T1)
        k_gemm_rows_mo_build<true><<<grid, block, 0, s>>>((const uint8_t *)tiled, (int)rows, (int)(cols / 32), tiled_row_bytes(dtype, cols / 64), (uint8_t *)mo);
    else return ifa_fail(IFA_ERR_ARG, "gemm_rows_mo_build: format %d has no MO layout (value = q * scale + base nibble formats only)", dtype);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

} // namespace ifa
