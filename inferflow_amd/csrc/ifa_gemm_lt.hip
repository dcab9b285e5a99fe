// ifa_gemm_lt.hip -- large-T linear layer: dequantise once, then the library's F16 GEMM.
//
// Where the prompt is long enough that the layer is MFMA-bound (T > 128 by default: past the split-K kernel of ifa_gemm.hip) the weight traffic of a full
// F16 copy is noise (4096x4096: 10 MB read + 33 MB written, ~10 us, against a 35-90 us GEMM) and hipBLASLt's
// 256x256-tile kernels reach 0.9-1.2 PFLOP/s where the fused dequantise-in-registers kernel of ifa_gemm.hip
// reaches 0.4-0.6 (profiles/r01_gemm_mfma_microbench.log, tools/probes/lib_gemm_probe.py).  This is the
// reference's own decomposition (TensorOpr::Dequantize + cublasGemmEx, inference_worker.cc:2374-2415) minus its
// transpose: Y[T][N] (row-major) is the column-major N x T product W (N x K, "T" operand, ld K) . X^T (K x T, ld K).
// Same arithmetic: weights rounded to half exactly like the dequant tensor, F16 x F16 products, fp32 accumulate,
// one F16 rounding, bias as a half add afterwards.  Below the threshold the fused kernel stays (weight-stream bound).
//
// hipBLASLt is bound at run time (dlopen) so that libinferflow_amd.so itself links only the HIP runtime; when the
// library is absent ifa_gemm keeps using its own kernel for every T.
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <tuple>
#include <hipblaslt/hipblaslt.h>
#include "ifa_host.h"
#include "ifa_codec.h"

namespace ifa {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Q4_B32T1A/B -> F16, coalesced: 4 lanes per 20-byte block, 8 values (one 16-byte store) each
__global__ void __launch_bounds__(256) k_dequant_q4_f16(const uint32_t *__restrict__ W, half_t *__restrict__ out, size_t nblocks)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t b = gid >> 2;
    if (b >= nblocks) return;
    const int part = (int)(gid & 3);
    const uint32_t sb = W[b * 5], c = W[b * 5 + 1 + part];
    const float base = hbits2f((uint16_t)(sb & 0xFFFFu)), scale = hbits2f((uint16_t)(sb >> 16));
    const uint32_t lo = c & 0x0F0F0F0Fu, hi = (c >> 4) & 0x0F0F0F0Fu;         // value 2b = low nibble of byte b, 2b+1 = its high nibble
    const float ql[4] = {ubyte_f32<0>(lo), ubyte_f32<1>(lo), ubyte_f32<2>(lo), ubyte_f32<3>(lo)};
    const float qh[4] = {ubyte_f32<0>(hi), ubyte_f32<1>(hi), ubyte_f32<2>(hi), ubyte_f32<3>(hi)};
    half_t v[8];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        v[2 * b] = f2h(__builtin_fmaf(ql[b], scale, base));                     // q*scale exact: fma == mul + add
        v[2 * b + 1] = f2h(__builtin_fmaf(qh[b], scale, base));
    }
    *reinterpret_cast<u32x4 *>(out + gid * 8) = *reinterpret_cast<const u32x4 *>(v);
}

__global__ void __launch_bounds__(256) k_add_bias_rows(half_t *__restrict__ Y, const half_t *__restrict__ bias, size_t total, int N)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) Y[i] = f2h(h2f(Y[i]) + h2f(bias[i % (size_t)N]));
}

struct LtApi {
    bool ok = false;
    decltype(&hipblasLtCreate) Create = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
    decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
    decltype(&hipblasLtMatmul) Matmul = nullptr;
    decltype(&hipblasLtDestroy) Destroy = nullptr;                                  // optional: teardown only
    decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
};

static const LtApi &lt_api()
{
    static LtApi api = [] {
        LtApi a;
        void *so = nullptr;
        for (const char *name : {"libhipblaslt.so.1", "libhipblaslt.so", "/opt/rocm/lib/libhipblaslt.so.1"})
            if ((so = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!so) return a;
#define IFA_LT_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(so, #sym))
        IFA_LT_SYM(Create, hipblasLtCreate); IFA_LT_SYM(LayoutCreate, hipblasLtMatrixLayoutCreate);
        IFA_LT_SYM(DescCreate, hipblasLtMatmulDescCreate); IFA_LT_SYM(DescSet, hipblasLtMatmulDescSetAttribute);
        IFA_LT_SYM(PrefCreate, hipblasLtMatmulPreferenceCreate); IFA_LT_SYM(PrefSet, hipblasLtMatmulPreferenceSetAttribute);
        IFA_LT_SYM(Heuristic, hipblasLtMatmulAlgoGetHeuristic); IFA_LT_SYM(Matmul, hipblasLtMatmul);
        IFA_LT_SYM(Destroy, hipblasLtDestroy); IFA_LT_SYM(DescDestroy, hipblasLtMatmulDescDestroy); IFA_LT_SYM(LayoutDestroy, hipblasLtMatrixLayoutDestroy);
        IFA_LT_SYM(PrefDestroy, hipblasLtMatmulPreferenceDestroy);
#undef IFA_LT_SYM
        a.ok = a.Create && a.LayoutCreate && a.DescCreate && a.DescSet && a.PrefCreate && a.PrefSet && a.Heuristic && a.Matmul;
        return a;
    }();
    return api;
}

struct LtPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t workspace = 0;
    bool ok = false;
};

// one context per (device, stream): the F16 scratch copy of the weights is reused call after call on that stream
struct LtContext {
    hipblasLtHandle_t handle = nullptr;
    void *workspace = nullptr;
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    std::map<std::tuple<size_t, size_t, size_t>, LtPlan> plans;
};
constexpr size_t LT_WORKSPACE = (size_t)64 << 20;

static std::mutex g_lt_mutex;
static std::map<std::pair<int, hipStream_t>, LtContext> g_lt_ctx;
static int g_lt_min_tokens = -1;          // -1: not initialised (environment IFA_GEMM_LT_MIN_TOKENS); 0: never -- the DEFAULT: the library route is opt-in,
                                          // the in-tree kernels (ifa_gemm.hip) serve every T unless it is switched on

static int lt_min_tokens()
{
    if (g_lt_min_tokens < 0) {
        const char *e = getenv("IFA_GEMM_LT_MIN_TOKENS");
        g_lt_min_tokens = e ? std::max(0, atoi(e)) : 0;
    }
    return g_lt_min_tokens;
}

bool gemm_lt_wanted(size_t tokens)
{
    const int mt = lt_min_tokens();
    return mt > 0 && tokens >= (size_t)mt && lt_api().ok;
}

// IFA_OK, or IFA_ERR_STATE when the library cannot take this problem (the caller then runs its own kernel)
int gemm_lt(int w_dtype, const void *W, size_t N, size_t K, const void *X, size_t T, const void *bias, void *Y, hipStream_t s)
{
    const LtApi &api = lt_api();
    if (!api.ok) return IFA_ERR_STATE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return IFA_ERR_STATE;
    std::lock_guard<std::mutex> lock(g_lt_mutex);
    LtContext &ctx = g_lt_ctx[{dev, s}];
    if (!ctx.handle) {
        if (api.Create(&ctx.handle) != HIPBLAS_STATUS_SUCCESS) { ctx.handle = nullptr; return IFA_ERR_STATE; }
        if (hipMalloc(&ctx.workspace, LT_WORKSPACE) != hipSuccess) { ctx.workspace = nullptr; return IFA_ERR_STATE; }
    }
    const void *Wh = W;
    if (w_dtype != F16) {
        const size_t need = N * K * 2;
        if (need > ctx.scratch_bytes) {
            if (ctx.scratch) { (void)hipStreamSynchronize(s); (void)hipFree(ctx.scratch); }
            ctx.scratch = nullptr; ctx.scratch_bytes = 0;
            if (hipMalloc(&ctx.scratch, need) != hipSuccess) return IFA_ERR_STATE;
            ctx.scratch_bytes = need;
        }
        if (w_dtype == Q4_B32T1A || w_dtype == Q4_B32T1B) {
            const size_t nblocks = N * (K / 32);
            k_dequant_q4_f16<<<dim3((unsigned)ifa_cdiv(nblocks * 4, 256)), dim3(256), 0, s>>>((const uint32_t *)W, (half_t *)ctx.scratch, nblocks);
        } else if (ifa_dequantize(w_dtype, W, N, K, ctx.scratch, (ifa_stream)s) != IFA_OK) {
            return IFA_ERR_STATE;
        }
        Wh = ctx.scratch;
    }
    // a serving workload sees many prompt lengths: the plans of a stream are capped (all dropped when the cap is reached;
    // a plan costs a heuristic query to rebuild)
    constexpr size_t LT_MAX_PLANS = 256;
    if (ctx.plans.size() >= LT_MAX_PLANS && !ctx.plans.count(std::make_tuple(N, K, T))) {
        for (auto &kv : ctx.plans) {
            LtPlan &q = kv.second;
            if (q.ok && api.DescDestroy) (void)api.DescDestroy(q.desc);
            if (api.LayoutDestroy) { if (q.a) (void)api.LayoutDestroy(q.a); if (q.b) (void)api.LayoutDestroy(q.b); if (q.c) (void)api.LayoutDestroy(q.c); }
        }
        ctx.plans.clear();
    }
    LtPlan &p = ctx.plans[std::make_tuple(N, K, T)];
    if (!p.desc) {
        const int32_t op_t = HIPBLAS_OP_T, op_n = HIPBLAS_OP_N;
        hipblasLtMatmulPreference_t pref = nullptr;
        hipblasLtMatmulHeuristicResult_t res[1];
        int found = 0;
        const uint64_t ws = LT_WORKSPACE;
        p.ok = api.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS
            && api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_t, sizeof(op_t)) == HIPBLAS_STATUS_SUCCESS
            && api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_n, sizeof(op_n)) == HIPBLAS_STATUS_SUCCESS
            && api.LayoutCreate(&p.a, HIP_R_16F, K, N, (int64_t)K) == HIPBLAS_STATUS_SUCCESS       // W^T as stored: K x N, ld K
            && api.LayoutCreate(&p.b, HIP_R_16F, K, T, (int64_t)K) == HIPBLAS_STATUS_SUCCESS       // X^T: K x T, ld K
            && api.LayoutCreate(&p.c, HIP_R_16F, N, T, (int64_t)N) == HIPBLAS_STATUS_SUCCESS       // Y^T: N x T, ld N
            && api.PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS
            && api.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws)) == HIPBLAS_STATUS_SUCCESS
            && api.Heuristic(ctx.handle, p.desc, p.a, p.b, p.c, p.c, pref, 1, res, &found) == HIPBLAS_STATUS_SUCCESS
            && found > 0 && res[0].state == HIPBLAS_STATUS_SUCCESS && res[0].workspaceSize <= LT_WORKSPACE;
        if (p.ok) { p.algo = res[0].algo; p.workspace = res[0].workspaceSize; }
        if (pref && api.PrefDestroy) (void)api.PrefDestroy(pref);
        if (!p.desc) p.desc = reinterpret_cast<hipblasLtMatmulDesc_t>(1);      // remember the failure
    }
    if (!p.ok) return IFA_ERR_STATE;
    const float alpha = 1.0f, beta = 0.0f;
    if (api.Matmul(ctx.handle, p.desc, &alpha, Wh, p.a, X, p.b, &beta, Y, p.c, Y, p.c, &p.algo, ctx.workspace, p.workspace, s)
        != HIPBLAS_STATUS_SUCCESS)
        return IFA_ERR_STATE;
    if (bias) k_add_bias_rows<<<dim3((unsigned)ifa_cdiv(T * N, 256)), dim3(256), 0, s>>>((half_t *)Y, (const half_t *)bias, T * N, (int)N);
    return IFA_OK;
}

} // namespace ifa

namespace ifa { void attn_release_stream(int dev, hipStream_t s); }      // ifa_attn.hip: the score-tile workspace of the stream

extern "C" int ifa_gemm_release_stream(ifa_stream stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return IFA_OK;
    (void)hipStreamSynchronize(ifa_s(stream));
    ifa::attn_release_stream(dev, ifa_s(stream));
    std::lock_guard<std::mutex> lock(ifa::g_lt_mutex);
    auto it = ifa::g_lt_ctx.find({dev, ifa_s(stream)});
    if (it == ifa::g_lt_ctx.end()) return IFA_OK;
    (void)hipStreamSynchronize(ifa_s(stream));
    const ifa::LtApi &api = ifa::lt_api();
    for (auto &kv : it->second.plans) {
        ifa::LtPlan &p = kv.second;
        if (p.ok && api.DescDestroy) (void)api.DescDestroy(p.desc);
        if (api.LayoutDestroy) { if (p.a) (void)api.LayoutDestroy(p.a); if (p.b) (void)api.LayoutDestroy(p.b); if (p.c) (void)api.LayoutDestroy(p.c); }
    }
    if (it->second.handle && api.Destroy) (void)api.Destroy(it->second.handle);
    if (it->second.scratch) (void)hipFree(it->second.scratch);
    if (it->second.workspace) (void)hipFree(it->second.workspace);
    ifa::g_lt_ctx.erase(it);
    return IFA_OK;
}

extern "C" int ifa_gemm_library_available(void)
{
    return ifa::lt_api().ok ? 1 : 0;
}

extern "C" int ifa_gemm_library_min_tokens(int min_tokens)
{
    std::lock_guard<std::mutex> lock(ifa::g_lt_mutex);
    const int prev = ifa::lt_min_tokens();
    if (min_tokens >= 0) ifa::g_lt_min_tokens = min_tokens;
    return ifa::lt_api().ok ? prev : 0;
}
