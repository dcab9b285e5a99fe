// ifa_gemm_rows_mo.hip -- the 2..16-row weight-streaming GEMM (ifa_gemm_rows_mfma.hip) on weights in the MO layout
// ("MFMA operand order", ifa_gemm_rows_mfma.h): the instantiations and their launcher.  A lane's 16-byte request is its own
// A operand, a superstep of a tile one contiguous KiB per wave; 16 activation rows x 4096 columns fit the LDS (no patches), so
// wq | wk | wv, wo and w1 / w3 of a 4096-wide model are single-chunk kernels (CH = 1: straight-line, counted waits) for every
// batch of 2..16 rows; longer rows (w2) walk 4096-column chunks.  Rows are staged as 8 or 16 (TX).
#include <mutex>
#include "ifa_gemm_rows_mfma_body.h"

namespace ifa {

static int dec_num_cus_rows()
{
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t prop; ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256; }
    return ncu;
}
// per-(device, stream) scratch of the K-parts launches: the parts' tile sums as 8-byte {tag, value} granules, zero between launches
struct KPartsScratch { void *p = nullptr; size_t bytes = 0; };
static std::mutex g_kparts_mu;
static std::map<std::pair<int, hipStream_t>, KPartsScratch> g_kparts;
static const int g_rows_kparts_off = getenv("IFA_ROWS_KPARTS_OFF") ? 1 : 0;       // tuning aid (A / B)
// (from two chunks on: wo of 17..32 queries as two parts 10.5 -> 9.5 us; 32 queries 3.22 -> 3.18 ms)
static const int KPARTS_MIN_CHUNKS = getenv("IFA_ROWS_KPARTS_MIN") ? atoi(getenv("IFA_ROWS_KPARTS_MIN")) : 2;
static int rows_kparts_scratch(hipStream_t s, size_t need, void **out)
{
    int dev = 0;
    IFA_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_kparts_mu);
    KPartsScratch &sc = g_kparts[std::make_pair(dev, s)];
    if (sc.bytes < need) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &st);
        if (st != hipStreamCaptureStatusNone) return ifa_fail(IFA_ERR_STATE, "rows GEMM K parts: the scratch must exist before a capture (gemm_rows_kparts_reserve)");
        IFA_HIP_CHECK(hipStreamSynchronize(s));
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr; sc.bytes = 0;
        const size_t alloc = std::max(need, (size_t)16 << 20);
        IFA_HIP_CHECK(hipMalloc(&sc.p, alloc));
        IFA_HIP_CHECK(hipMemsetAsync(sc.p, 0, alloc, s));
        sc.bytes = alloc;
    }
    *out = sc.p;
    return IFA_OK;
}
// the scratch of the stream exists (>= 16 MB: every 7B..70B-width product of up to 32 rows) -- called before a capture
int gemm_rows_kparts_reserve(hipStream_t s)
{
    void *p = nullptr;
    (void)wait_err_word();      // the device's wait-error word is pinned host memory: allocated HERE, never under a stream capture
                                // (hipHostMalloc inside a thread-local capture fails and poisons it: "operation failed due to a
                                //  previous error during capture" at the first captured K-parts launch, r05 bench_batch at 17 queries)
    return rows_kparts_scratch(s, 1, &p);       // (any size: the allocation is >= 16 MB)
}
void rows_kparts_release(int dev, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_kparts_mu);
    auto it = g_kparts.find(std::make_pair(dev, s));
    if (it == g_kparts.end()) return;
    if (it->second.p) (void)hipFree(it->second.p);
    g_kparts.erase(it);
}

template <int MT, int KPM>
static int mo_launch_kparts(int wgs, size_t smem, const GmArgs &P, int epi, hipStream_t s)
{
    // returns 1 (nothing launched) when the grid cannot be resident at once: the caller takes the launch without K parts
#define IFA_KP(TXV, EPIV) { auto kern = k_gemm_rows_mfma<MT, TXV, EPIV, 0, true, 0, KPM>; \
        if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        if (!wait_grid_fits((const void *)kern, GM_THREADS, smem, wgs)) return 1; \
        kern<<<dim3((unsigned)wgs), dim3(GM_THREADS), smem, s>>>(P); }
    if (P.T <= 16) { if (epi == GM_PLAIN) IFA_KP(16, GM_PLAIN) else IFA_KP(16, GM_RESIDUAL) }
    else { if (epi == GM_PLAIN) IFA_KP(32, GM_PLAIN) else IFA_KP(32, GM_RESIDUAL) }
#undef IFA_KP
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

template <int MT, int TX, int EPI, int NORM, int CH>
static int mo_launch4(int wgs, size_t smem, const GmArgs &P, hipStream_t s)
{
    auto kern = k_gemm_rows_mfma<MT, TX, EPI, NORM, true, CH>;
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)wgs), dim3(GM_THREADS), smem, s>>>(P);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

template <int MT, int TX>
static int mo_launch2(int wgs, size_t smem, const GmArgs &P, int epi, int norm, bool one, hipStream_t s)
{
    if (one) {
        if (epi == GM_PLAIN && norm == 0) return mo_launch4<MT, TX, GM_PLAIN, 0, 1>(wgs, smem, P, s);
        if (epi == GM_RESIDUAL && norm == 0) return mo_launch4<MT, TX, GM_RESIDUAL, 0, 1>(wgs, smem, P, s);
        if (epi == GM_PLAIN && norm == 1) return mo_launch4<MT, TX, GM_PLAIN, 1, 1>(wgs, smem, P, s);
        if constexpr (MT % 2 == 0) {
            if (epi == GM_GLU && norm == 1) return mo_launch4<MT, TX, GM_GLU, 1, 1>(wgs, smem, P, s);
            if (epi == GM_GLU && norm == 0) return mo_launch4<MT, TX, GM_GLU, 0, 1>(wgs, smem, P, s);
        }
    } else {
        if (epi == GM_PLAIN && norm == 0) return mo_launch4<MT, TX, GM_PLAIN, 0, 0>(wgs, smem, P, s);
        if (epi == GM_RESIDUAL && norm == 0) return mo_launch4<MT, TX, GM_RESIDUAL, 0, 0>(wgs, smem, P, s);
        if constexpr (MT % 2 == 0) { if (epi == GM_GLU && norm == 0) return mo_launch4<MT, TX, GM_GLU, 0, 0>(wgs, smem, P, s); }
    }
    return ifa_fail(IFA_ERR_ARG, "rows GEMM (MO): no kernel for epilogue %d / norm %d / %s", epi, norm, one ? "one chunk" : "chunked");
}

// 17..32 rows: two column tiles per A operand, rows staged in 2048-column chunks (always the chunk loop, no norm prologue)
template <int MT>
static int mo_launch32(int wgs, size_t smem, const GmArgs &P, int epi, int norm, hipStream_t s)
{
    if (epi == GM_PLAIN && norm == 0) return mo_launch4<MT, 32, GM_PLAIN, 0, 0>(wgs, smem, P, s);
    if (epi == GM_RESIDUAL && norm == 0) return mo_launch4<MT, 32, GM_RESIDUAL, 0, 0>(wgs, smem, P, s);
    if constexpr (MT % 2 == 0) { if (epi == GM_GLU && norm == 0) return mo_launch4<MT, 32, GM_GLU, 0, 0>(wgs, smem, P, s); }
    return ifa_fail(IFA_ERR_ARG, "rows GEMM (MO, 17..32 rows): no kernel for epilogue %d / norm %d", epi, norm);
}

template <int MT>
static int mo_launch1(int wgs, size_t smem, const GmArgs &P, int epi, int norm, bool one, hipStream_t s)
{
    // rows staged: 2, 4, 8 or 16 (staging eight rows for two queries was 1.7 us of norm arithmetic and LDS stores on duplicates)
    if (P.T <= 2) return mo_launch2<MT, 2>(wgs, smem, P, epi, norm, one, s);
    if (P.T <= 4) return mo_launch2<MT, 4>(wgs, smem, P, epi, norm, one, s);
    if (P.T <= 8) return mo_launch2<MT, 8>(wgs, smem, P, epi, norm, one, s);
    if (P.T <= 16) return mo_launch2<MT, 16>(wgs, smem, P, epi, norm, one, s);
    return mo_launch32<MT>(wgs, smem, P, epi, norm, s);
}

// 2..4 rows longer than one chunk, no norm: staged whole (CH = 2)
template <int MT>
static int mo_launch_whole(int wgs, size_t smem, const GmArgs &P, int epi, hipStream_t s)
{
    if (P.T <= 2) {
        if (epi == GM_PLAIN) return mo_launch4<MT, 2, GM_PLAIN, 0, 2>(wgs, smem, P, s);
        if (epi == GM_RESIDUAL) return mo_launch4<MT, 2, GM_RESIDUAL, 0, 2>(wgs, smem, P, s);
    } else {
        if (epi == GM_PLAIN) return mo_launch4<MT, 4, GM_PLAIN, 0, 2>(wgs, smem, P, s);
        if (epi == GM_RESIDUAL) return mo_launch4<MT, 4, GM_RESIDUAL, 0, 2>(wgs, smem, P, s);
    }
    return ifa_fail(IFA_ERR_ARG, "rows GEMM (MO, whole rows): no kernel for epilogue %d", epi);
}

int gemm_rows_mo_launch(const GmArgs &P, int epi, int norm, int wgs, int maxt, hipStream_t s)
{
    const bool one = P.nblk * 32 <= GmGeo<32>::CHUNK_COLS && P.T <= 16;
    const int K = P.nblk * 32;
    if (!one && norm == 0 && P.T <= 4 && K <= 16384 && (epi == GM_PLAIN || epi == GM_RESIDUAL) && gm_smem_whole(P.T <= 2 ? 2 : 4, K, maxt) <= (size_t)150 * 1024) {
        const size_t smw = gm_smem_whole(P.T <= 2 ? 2 : 4, K, maxt);
        switch (maxt) {
        case 1: return mo_launch_whole<1>(wgs, smw, P, epi, s);
        case 2: return mo_launch_whole<2>(wgs, smw, P, epi, s);
        case 3: return mo_launch_whole<3>(wgs, smw, P, epi, s);
        case 4: return mo_launch_whole<4>(wgs, smw, P, epi, s);
        case 6: return mo_launch_whole<6>(wgs, smw, P, epi, s);
        default: return mo_launch_whole<8>(wgs, smw, P, epi, s);
        }
    }
    if (norm == 1 && !one) return ifa_fail(IFA_ERR_ARG, "rows GEMM (MO): the norm prologue needs the whole row in one chunk");
    // K parts (GmArgs::kparts): 9..32 rows walking two chunks or more (one tile per workgroup before: maxt 1 or 2).  Every workgroup stages T rows of EVERY chunk for its
    // tiles: at 32 queries w2 (256 tiles, K = 11008: six 2048-column chunks) moved 180 MB of activations L2 -> LDS for 25 MB of
    // weights and ran at the L2's ~10 TB/s (21 us, rows-trace).  With kparts = the chunk count a workgroup takes ONE chunk of kparts
    // times as many tiles (rows staged once: 33 MB in all) and finishes its share of them (reduce-scatter of the fp32 tile sums).
    if (!one && norm == 0 && (epi == GM_PLAIN || epi == GM_RESIDUAL) && P.T >= 9 && !g_rows_kparts_off && !P.no_waits && waits_enabled() && (maxt == 1 || maxt == 2)) {
        const int chunk_sup = P.T <= 16 ? 32 : 16, nsup = P.nblk / 4, nchunk = (nsup + chunk_sup - 1) / chunk_sup;
        int kparts = 0;
        // (every workgroup of a group must be resident at once -- the finishers poll their group's other parts: a workgroup that
        //  waits for a free CU delays its whole group by a kernel's length (w2, 258 workgroups: median end 14.6 us, the last group 27))
        const int ntiles_all = (P.total_rows + 15) / 16;
        for (int d = std::min(nchunk, 8 / maxt); d >= 2 && nchunk >= KPARTS_MIN_CHUNKS; d--) {
            if (nchunk % d != 0 || maxt * d == 5 || maxt * d == 7) continue;
            if ((ntiles_all + maxt * d - 1) / (maxt * d) * d > std::min(dec_num_cus_rows(), visible_cus())) continue;
            kparts = d; break;
        }
        if (kparts >= 2) {
            const int ntiles = (P.total_rows + 15) / 16, maxt_k = maxt * kparts, groups = (ntiles + maxt_k - 1) / maxt_k;
            const int NTc = P.T > 16 ? 2 : 1;
            GmArgs Q = P;
            void *scratch = nullptr;
            int rc = rows_kparts_scratch(s, (size_t)groups * maxt_k * kparts * NTc * 256 * 8, &scratch);
            if (rc) return rc;
            Q.kparts = kparts; Q.kpart_sums = (unsigned long long *)scratch; Q.wait_err = wait_err_word();
            const size_t smem_k = gm_smem(P.T, maxt_k, 1);
            const int wgs_k = groups * kparts;
            int rk = 1;        // 1: not launched (no instance for this shape, or the grid cannot be resident at once) -> the plain launch below
            if (maxt == 1) switch (maxt_k) {
                case 2: rk = mo_launch_kparts<2, 1>(wgs_k, smem_k, Q, epi, s); break;
                case 3: rk = mo_launch_kparts<3, 1>(wgs_k, smem_k, Q, epi, s); break;
                case 4: rk = mo_launch_kparts<4, 1>(wgs_k, smem_k, Q, epi, s); break;
                case 6: rk = mo_launch_kparts<6, 1>(wgs_k, smem_k, Q, epi, s); break;
                case 8: rk = mo_launch_kparts<8, 1>(wgs_k, smem_k, Q, epi, s); break;
                default: break;
            } else switch (maxt_k) {
                case 4: rk = mo_launch_kparts<4, 2>(wgs_k, smem_k, Q, epi, s); break;
                case 6: rk = mo_launch_kparts<6, 2>(wgs_k, smem_k, Q, epi, s); break;
                case 8: rk = mo_launch_kparts<8, 2>(wgs_k, smem_k, Q, epi, s); break;
                default: break;
            }
            if (rk != 1) return rk;
        }
    }
    const size_t smem = gm_smem(P.T, maxt, 1);
    switch (maxt) {
    case 1: return mo_launch1<1>(wgs, smem, P, epi, norm, one, s);
    case 2: return mo_launch1<2>(wgs, smem, P, epi, norm, one, s);
    case 3: return mo_launch1<3>(wgs, smem, P, epi, norm, one, s);
    case 4: return mo_launch1<4>(wgs, smem, P, epi, norm, one, s);
    case 6: return mo_launch1<6>(wgs, smem, P, epi, norm, one, s);
    default: return mo_launch1<8>(wgs, smem, P, epi, norm, one, s);
    }
}

// Mixture of experts, experts with 2..8 rows ("smalls", ifa_moe.h): blockIdx.y is one expert's group of consecutive rows of the
// gathered activations; its MO weights come from the pointer table {w1, w3, w2, -}.  EPI == GM_GLU: w1 and w3 of the expert in
// ONE launch with act(w1 x) * (w3 x) as the output (the tiled path runs two launches and an element-wise kernel).
template <int MAXT, int EPI, int CH>
__global__ void __launch_bounds__(GM_THREADS) k_gemm_rows_mo_grouped(const MoeSmallGroup grp, int rows, int nblk, int act_kind, const half_t *__restrict__ X,
                                                                     half_t *__restrict__ Y, int whole_lds)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.y >= grp.counts[3]) return;
    const MoeTile gq = grp.smalls[blockIdx.y];
    GmArgs P;
    P.W[0] = grp.wtab_mo[4 * gq.expert + grp.which_tiled]; P.W[1] = nullptr; P.W[2] = nullptr;
    P.W1 = EPI == GM_GLU ? grp.wtab_mo[4 * gq.expert + grp.which_tiled + 1] : nullptr;
    P.rows[0] = rows; P.rows[1] = 0; P.rows[2] = 0; P.nsets = 1; P.total_rows = rows; P.nblk = nblk; P.T = gq.nrows;
    P.X = X + (size_t)gq.row0 * nblk * 32; P.ldx = nblk * 32; P.multi_base = 0.0f; P.eps = 0.0f; P.norm_w = nullptr;
    P.bias[0] = nullptr; P.bias[1] = nullptr; P.bias[2] = nullptr; P.bias1 = nullptr;
    P.Y = Y + (size_t)gq.row0 * rows; P.res = nullptr; P.ldy = rows; P.ldres = 0; P.act_kind = act_kind;
    P.Yset[0] = nullptr; P.Yset[1] = nullptr; P.Yset[2] = nullptr; P.ldyset[0] = 0; P.ldyset[1] = 0; P.ldyset[2] = 0;
    P.mo = 1; P.trace = nullptr; P.kparts = 0; P.kpart_sums = nullptr;
    // groups of 2..4 rows (most of them: 16 entries over 8 experts): four rows staged instead of eight, and rows longer than a chunk
    // (w2) staged whole -- no re-staging, no barriers in the chunk loop
    if (gq.nrows <= 4 && (CH == 1 || (size_t)4 * ((size_t)nblk * 64 + 16) <= (size_t)whole_lds)) {
        if constexpr (CH == 1) gemm_rows_mfma_body<MAXT, 4, EPI, 0, true, 1>(P, smem);
        else gemm_rows_mfma_body<MAXT, 4, EPI, 0, true, 2>(P, smem);
    } else gemm_rows_mfma_body<MAXT, 8, EPI, 0, true, CH>(P, smem);
}

template <int MT, int EPI, int CH>
static int mo_launch_grouped(int wgs, int groups, size_t smem, const MoeSmallGroup &grp, size_t rows, size_t cols, int act_kind, const void *X, void *Y, hipStream_t s)
{
    auto kern = k_gemm_rows_mo_grouped<MT, EPI, CH>;
    // rows longer than a chunk: room for four whole rows when that fits (groups of <= 4 rows then skip the chunk re-staging)
    int whole_lds = 0;
    if (CH == 0 && gm_smem_whole(4, (int)cols, MT) <= (size_t)150 * 1024) { whole_lds = (int)gm_smem_whole(4, (int)cols, MT); smem = std::max(smem, (size_t)whole_lds); }
    if (smem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<dim3((unsigned)wgs, (unsigned)groups), dim3(GM_THREADS), smem, s>>>(grp, (int)rows, (int)(cols / 32), act_kind, (const half_t *)X, (half_t *)Y, whole_lds);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// rows / cols of ONE expert matrix (glu: of w1 and of w3); X / Y: the gathered activations / outputs of all entries; groups of 2..8 rows
int gemm_rows_mo_grouped(const MoeSmallGroup &grp, size_t rows, size_t cols, const void *X, void *Y, int max_groups, int glu, int act_kind, hipStream_t s)
{
    if (!grp.wtab_mo || cols % 128 != 0 || rows == 0) return ifa_fail(IFA_ERR_STATE, "grouped rows GEMM (MO): %zu x %zu", rows, cols);
    if (max_groups <= 0) return IFA_OK;
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t prop; ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256; }
    // the experts share the chip: two workgroups fit a CU (66 KB of LDS each), so up to 2 * CUs / groups workgroups per expert
    const int ntiles = (int)((rows + 15) / 16), cap = std::max(32, 2 * ncu / max_groups);
    int w = std::min(cap, ntiles), mt = (ntiles + w - 1) / w;
    if (glu) {                          // pairs (w1 tile, w3 tile): 1..4 pairs per workgroup
        if (mt > 4) { mt = 4; w = (ntiles + 3) / 4; }
        mt = mt == 3 ? 6 : mt * 2;
    } else {
        if (mt > 8) { mt = 8; w = (ntiles + 7) / 8; }
        if (mt == 5) mt = 6;
        if (mt == 7) mt = 8;
    }
    const bool one = cols <= (size_t)GmGeo<32>::CHUNK_COLS;
    const size_t smem = gm_smem(8, mt, 1);
#define IFA_MOG(MTV) \
    if (mt == MTV) { \
        if (glu) { if constexpr (MTV % 2 == 0) return one ? mo_launch_grouped<MTV, GM_GLU, 1>(w, max_groups, smem, grp, rows, cols, act_kind, X, Y, s) \
                                                          : mo_launch_grouped<MTV, GM_GLU, 0>(w, max_groups, smem, grp, rows, cols, act_kind, X, Y, s); } \
        else return one ? mo_launch_grouped<MTV, GM_PLAIN, 1>(w, max_groups, smem, grp, rows, cols, act_kind, X, Y, s) \
                        : mo_launch_grouped<MTV, GM_PLAIN, 0>(w, max_groups, smem, grp, rows, cols, act_kind, X, Y, s); \
    }
    IFA_MOG(1) IFA_MOG(2) IFA_MOG(3) IFA_MOG(4) IFA_MOG(6) IFA_MOG(8)
#undef IFA_MOG
    return ifa_fail(IFA_ERR_ARG, "grouped rows GEMM (MO): %d tiles per workgroup", mt);
}

} // namespace ifa
