// ifa_engine.hip -- per-device decode worker: the MI355X counterpart of
// GpuInferenceWorker (src/transformer/inference_worker.cc:234-340, :762-981,
// :983-1405, :1726-1903, :552-624) for one query at a time.
//
//  * ifa_model_forward(): any number of new tokens, op-by-op through the same
//    C-ABI ops the reference worker would call (TensorOpr / TensorMul
//    counterparts), host-driven, synchronous on return.  Used for prefill and
//    as the "unfused" cross-check of the decode kernels.
//  * ifa_model_decode(): batch-1 greedy decode with the fused kernels of
//    ifa_decode_kernels.h, one hipGraph replay per token, token fed back on the
//    device (no host round trip inside a batch of steps).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "ifa_host.h"
#include "ifa_decode_kernels.h"
#include "ifa_decode_attn.h"
#include "ifa_decode_gemv.h"
#include "ifa_moe.h"
#include "ifa_gemm_rows_mfma.h"
#include "ifa_gemm_big.h"
#include "ifa_decode_lmhead_tail.h"
#include "ifa_decode_singles.h"
#include "ifa_decode_qkv_attn.h"
#include "ifa_decode_chain.h"

using namespace ifa;

namespace {

enum { T_EMBD = 0, T_OUT_NORM = 1, T_OUT_NORM_B = 2, T_LM_HEAD = 3,
       T_ATTN_NORM = 10, T_ATTN_NORM_B = 11, T_WQ = 12, T_WK = 13, T_WV = 14, T_WO = 15,
       T_FFN_NORM = 16, T_FFN_NORM_B = 17, T_W1 = 18, T_W2 = 19, T_W3 = 20, T_MOE_GATE = 21,
       T_WQ_B = 22, T_WK_B = 23, T_WV_B = 24, T_WO_B = 25, T_W1_B = 26, T_W2_B = 27, T_W3_B = 28,
       T_ATTN_POST_NORM = 29, T_ATTN_POST_NORM_B = 30, T_FFN_POST_NORM = 31, T_FFN_POST_NORM_B = 32,      // self_attn.post_norm / feed_forward.post_norm (model.h:168-276)
       T_MAX = 36 };

struct Tensor {
    int dtype = -1;
    void *data = nullptr;    // reference layout (AoS blocks / F16), engine-owned
    void *tiled = nullptr;   // row-local plane layout for the fused kernels (or null)
    void *mo = nullptr;      // MFMA-operand-order copy for the small-batch rows GEMM (ifa_gemm_rows_mfma.h), built on first use
    void *x32 = nullptr;     // 64-weight nibble formats: the same values as Q4_B32T1A reference-layout blocks, for the large-tile prefill GEMM
    size_t rows = 0, cols = 0;
    bool present() const { return data != nullptr; }
};

struct Layer {
    Tensor t[T_MAX];
    std::vector<Tensor> experts;   // [expert][3]: w1, w2, w3 (MoE layers)
    void *moe_table = nullptr;     // device: [expert][4] tiled pointers {w1, w3, w2, -} for the fused decode kernels
    void *moe_table_aos = nullptr; // device: [expert][3] reference-layout pointers {w1, w2, w3} for the grouped T > 1 launches
    void *moe_table_mo = nullptr;  // device: [expert][4] MO copies {w1, w3, w2, -} for the experts with 2..8 rows of a batched step (ensure_mo)
    void *kcache = nullptr, *vcache = nullptr;
};

} // namespace

struct ifa_model {
    ifa_model_config cfg;
    std::vector<Layer> layers;
    Tensor g[10];
    hipStream_t stream = nullptr;
    // side stream + fork / join events of the batched MoE step: the single-row experts run next to the small groups (both stream
    // expert matrices nobody else reads and neither saturates the memory system alone); created on first use
    hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; int opt_moe_overlap = 1;
    bool own_stream = true;
    bool finalized = false;
    // scratch
    half_t *x = nullptr, *x2 = nullptr, *xn = nullptr, *hn = nullptr, *q = nullptr, *k = nullptr, *v = nullptr;
    half_t *dqkv = nullptr;     // the decode step's q | k | v vector as ONE buffer (k_dec_attn addresses k and v from q's pointer)
    half_t *bqkv = nullptr;     // fused batched step: [queries][q | k | v]
    float *brope = nullptr;     // fused batched step: [queries][head_dim] (cos, sin) pairs
    size_t bqkv_rows = 0;
    half_t *att = nullptr, *a = nullptr, *f = nullptr, *t1 = nullptr, *t2 = nullptr, *logits = nullptr;
    uint8_t *xq = nullptr;
    int8_t *attq = nullptr;        // XqImage of the attention output (Q8_B32T2), written by the fused attention kernels for the Wo GEMV
    half_t *moe_gate = nullptr, *moe_out = nullptr;   // MoE: router probabilities [T][experts], one expert's output rows
    half_t *moe_in = nullptr, *moe_wdev = nullptr;    // MoE: one expert's gathered input rows; per-row weights
    int *moe_route = nullptr;                          // device: fused decode routing, [0..7] expert ids, halfs at byte 32: weights
    int *moe_idx = nullptr, *moe_pin = nullptr;        // MoE: row lists of all experts, back to back (device / pinned staging)
    // MoE over T > 1 rows without the host (moe_ffn_device): routing, lists, gathered rows of ALL experts at once
    int *moe_sel = nullptr, *moe_epos = nullptr, *moe_counts = nullptr;
    half_t *moe_selw = nullptr, *moe_g1 = nullptr, *moe_g3 = nullptr, *moe_gin = nullptr, *moe_gout = nullptr;
    uint8_t *moe_xq_in = nullptr, *moe_xq_mid = nullptr;
    void *moe_tiles = nullptr, *moe_singles = nullptr, *moe_smalls = nullptr;
    int opt_moe_device = 1;
    int *state = nullptr;          // device: see k_dec_gather
    float *rope_tab = nullptr;     // device: [head_dim/2][2]
    long long *trace = nullptr;    // device: [2048][8] optional kernel phase stamps
    int opt_trace = 0, opt_bench_mode = 0, opt_touch_stride = 65536;
    int *tokens_dev = nullptr;
    int *host_pinned = nullptr;    // pinned staging for state / tokens
    int scratch_tokens = 0;
    size_t kv_row_bytes = 0;
    // decode graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    // the same step captured opt_graph_steps times in a row (option graph_steps, default 1 = off): a replay boundary costs ~8.6 us
    // against ~1.5 us between two launches inside a graph (rocprofv3 trace of the bench) -- but replays of 4 / 8 / 16 steps measured
    // SLOWER per token than single steps (1.2514 / 1.2524 / 1.2502 ms against 1.2427: profiles/r04_ab_options.log), so it stays opt-in
    hipGraph_t graph_n = nullptr; hipGraphExec_t graph_exec_n = nullptr; int graph_n_steps = 0, opt_graph_steps = 1;
    // tensor parallelism: a seam's "layer input + merged product (+ bias)" waiting to be formed in the prologue of the
    // GEMV that consumes it (instead of one or two tiny add kernels per seam)
    struct PendingAdd { const half_t *x = nullptr, *add = nullptr, *bias = nullptr; half_t *out = nullptr; bool on = false; } pend;
    int opt_tp_fuse_add = 1;
    // dynamic batching tables (forward_batch)
    void *batch_tab_dev = nullptr, *batch_tab_pin = nullptr;
    size_t batch_tab_bytes = 0;
    std::map<int, hipGraphExec_t> batch_graphs;      // captured batched step per batch size (dense models)
    // long-context decode attention (keys split over workgroups): workspace, switch and the context it starts at
    DecAttnSplitWs attn_ws = {nullptr, nullptr, nullptr, 8};
    // attention as the tail of the QKV launch (ifa_decode_qkv_attn.h): granules [layers][(heads + 2 kv_heads) * head_dim], the
    // decode-call counter the tags are built from, its own error word
    int opt_fuse_attn = 1, opt_fuse_attn_timeout_us = 20000, qa_on = 0, qa_gk = 0;
    unsigned long long *qa_gran = nullptr;
    unsigned *qa_call = nullptr, *qa_err = nullptr, qa_calls = 0;
    // consecutive GEMV ops of a layer as ONE launch with the next op's rows requested before the hand-off (ifa_decode_chain.h):
    // option fuse_ffn = 1: W1 | W3 -> W2; 2: Wo -> W1 | W3 -> W2.  ch_on = what the captured step uses.  Granules [layers][dim + ffn].
    int opt_fuse_ffn = 0, ch_on = 0, opt_chain_late_w2 = 0;
    uint32_t *ch_gran = nullptr, *ch_flags = nullptr;      // flags [layers][2][CH_FLAGS]
    // the end of the step as one launch (ifa_decode_lmhead_tail.h): lm_head + argmax + state advance + the next step's gather.
    // st_on = what the captured step uses (F16 lm_head with the RMS / no final norm)
    int opt_step_tail = 1, st_on = 0;
    unsigned long long *st_keys = nullptr; unsigned *st_counter = nullptr; int st_keys_n = 0;
    int attn_pb = 256, opt_attn_kt = 1;      // cache rows the one-workgroup decode attention requests at entry (64 / 128 / 256: the bucket the call stays inside); K rows through the LDS tile
    int attn_split = 0, opt_attn_split_ctx = 512, opt_batch_graph = 0, opt_gemm_rows = 1, opt_batch_fused = 1, opt_moe_router_fused = 1, opt_prefill_big = 1, opt_rows_mo = 1, opt_moe_singles = 1;
    // independent KV caches ("query slots", one per concurrent query like the reference's per-query
    // LayerKVCache sets): the inactive ones park their cache pointers and captured graph here
    struct KvSlot { std::vector<void *> k, v; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
    std::vector<KvSlot> slots;
    int cur_slot = 0;
    // options
    int opt_attn_q8 = 1;
    // multi-GPU decode driven from C (ifa_model_tp_decode): merge buffers, the distributed argmax's scratch, the captured step
    half_t *tp_a = nullptr, *tp_f = nullptr, *tp_hid = nullptr, *tp_logits = nullptr;
    float *tp_best = nullptr, *tp_gather = nullptr;
    int *tp_tok = nullptr;
    hipGraph_t tp_graph = nullptr;
    hipGraphExec_t tp_graph_exec = nullptr;
    // what the captured multi-GPU step was recorded with: communicator identities and every topology field its launches
    // depend on.  A call with anything else re-captures (a replay would use a stale communicator / offsets).
    struct TpKey {
        unsigned long long tp = 0, world = 0; int tp_size = 0, stage = 0, n_stages = 0, prev = 0, next = 0, src = 0, voff = 0, force = 0, fuse = 0, slot = 0, oneshot = 0;
        bool operator==(const TpKey &o) const { return tp == o.tp && world == o.world && tp_size == o.tp_size && stage == o.stage && n_stages == o.n_stages
                && prev == o.prev && next == o.next && src == o.src && voff == o.voff && force == o.force && fuse == o.fuse && slot == o.slot
                && oneshot == o.oneshot; }      // (oneshot: the captured collectives are the exchange or RCCL -- a switch forces a re-capture)
    } tp_key;
    const ifa_tp_topology *topo = nullptr;     // set by the partition entry points for the duration of a T > 1 / batched step
    size_t tp_rows_cap = 0;                    // rows the distributed-argmax scratch (tp_best / tp_gather / tp_tok) holds
    // is_attn_post_as_residual (model.h:113, default true): with an attention post-norm, the FFN's residual is the NORMALISED tensor
    int opt_attn_post_as_residual = 1;
    half_t *pn = nullptr;           // [tokens][dim] scratch of the post norms (allocated with the other activations)
    int opt_fused = 1, opt_graph = 1, opt_rpw_qkv = 0, opt_rpw_wo = 0, opt_rpw_ffn = 0, opt_rpw_w2 = 0, opt_rpw_lm = 0;
    int opt_debug_layers = 0;                  // > 0: the decode step runs only the first N layers (tools/debug_engine.py)
    // layer-wise parity tests (tests/test_gpu_layerwise_oracle.py): the fused decode step starts at layer debug_layer0 (with
    // debug_layers = N: layers [layer0, layer0 + N)) and, with debug_hidden_in, takes its input from the buffer "x" as the caller
    // left it instead of gathering the token's embedding row -- the SAME captured launches the bench times, fed the oracle's state
    int opt_debug_layer0 = 0, opt_debug_hidden_in = 0;
    // prompts ABOVE this many tokens take the four large-tile launches per layer (forward_ops, pf_big).  Round 4: 128.  Round 5: 47 -- with
    // four parts of K for the products that offer 32..96 tiles, 64 / 96 / 128 tokens run 5 / 8 / 9 % faster than the op-by-op layer
    // (7 products + 4 element-wise launches), 40 tokens the same (profiles/r05_prompt_lengths.log)
    int opt_prefill_big_min = 47;
    int opt_prefill_chunk = 1;      // prompts of 34..48 tokens as two passes of <= 32 tokens (ifa_model_forward)
    // round 6: prompts of prefill_big_min + 1 .. prefill_mid_max tokens take the four launches per layer from k_gemm_mid (ifa_gemm_mid.hip:
    // ring of direct-to-LDS stages, weights dequantised into the MFMA operand registers) when every linear is Q4_B32T1A / B
    int opt_prefill_mid = 1, opt_prefill_mid_max = 256;      // (320 tokens and up: the large tiles win again, profiles/r06_prefill_mid_ab.log)
    int opt_rows_kparts = 1, opt_gemm_splitk = 1;   // 0: never the launches whose workgroups wait for partner workgroups (K parts of the 9..32-row GEMM, split-K halves of the large-tile GEMM)
    int opt_debug_mo_alloc_fail = 0;           // tests: ensure_mo_build fails like an exhausted allocator after its first copy
    static constexpr int RING = 1024;
};

static void drop_graphs(ifa_model *m)
{
    if (m->tp_graph_exec) { (void)hipGraphExecDestroy(m->tp_graph_exec); m->tp_graph_exec = nullptr; }
    if (m->tp_graph) { (void)hipGraphDestroy(m->tp_graph); m->tp_graph = nullptr; }
    if (m->graph_exec) { (void)hipGraphExecDestroy(m->graph_exec); m->graph_exec = nullptr; }
    if (m->graph) { (void)hipGraphDestroy(m->graph); m->graph = nullptr; }
    if (m->graph_exec_n) { (void)hipGraphExecDestroy(m->graph_exec_n); m->graph_exec_n = nullptr; }
    if (m->graph_n) { (void)hipGraphDestroy(m->graph_n); m->graph_n = nullptr; }
    for (auto &sl : m->slots) {
        if (sl.exec) { (void)hipGraphExecDestroy(sl.exec); sl.exec = nullptr; }
        if (sl.graph) { (void)hipGraphDestroy(sl.graph); sl.graph = nullptr; }
    }
    for (auto &kv : m->batch_graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    m->batch_graphs.clear();
}

// keys split over workgroups past attn_split_ctx; 8 splits per head up to 2K keys, 16 up to 8K, 32 beyond (a head's K / V
// history streams through that many CUs: 8 splits left 16K-key contexts at 2 TB/s).  The captured steps hold the choice.
static void choose_attn_split(ifa_model *m, int reach)
{
    const int want = (m->opt_attn_split_ctx > 0 && reach > m->opt_attn_split_ctx) ? (reach > 8192 ? 32 : (reach > 2048 ? 16 : 8)) : 0;
    if (want != m->attn_split) { m->attn_split = want; drop_graphs(m); }
    // rows of the K / V cache the one-workgroup kernel requests before it knows the position: the bucket this call stays
    // inside (a longer context only costs the direct loads of the rows past it)
    const int pb = reach <= 64 ? 64 : (reach <= 128 ? 128 : 256);
    if (pb != m->attn_pb) { m->attn_pb = pb; drop_graphs(m); }
}

static void free_tensor(Tensor &t)
{
    if (t.data) (void)hipFree(t.data);
    if (t.tiled) (void)hipFree(t.tiled);
    if (t.mo) (void)hipFree(t.mo);
    if (t.x32) (void)hipFree(t.x32);
    t = Tensor();
}

static bool is_q4(int dt) { return dt == Q4_B32T1A || dt == Q4_B32T1B; }
// formats the MO copy of the rows GEMM takes: 4-bit codes with value q * scale + base (the 64-weight ones only through MO)
static bool rows_mo_fmt(int dt) { return is_q4(dt) || dt == Q4_B64T1 || dt == Q3H_B64T1; }
static int ensure_mo(ifa_model *m);
static int ensure_x32(ifa_model *m);
static const uint8_t *rows_w(const ifa_model *m, const Tensor &t);
static int rows_mo(const ifa_model *m, const Tensor &t);
static bool scale_on(float s) { return s < 0.9999f || s > 1.0001f; }     // the reference's test for "scale != 1"
// same tiled layout and arithmetic (the A/B variants differ only in how the quantizer picked base/scale)
static bool same_fmt(int a, int b) { return a == b || (is_q4(a) && is_q4(b)); }

// ------------------------------------------------------------------ dispatch
// reads one dword every `stride` bytes: warms the TLB / pulls lines towards L2+MALL
__global__ void __launch_bounds__(256) k_touch(const uint8_t *__restrict__ p, size_t bytes, size_t stride, int *sink)
{
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    int acc = 0;
    for (; i < bytes; i += (size_t)gridDim.x * blockDim.x * stride) acc += *reinterpret_cast<const int *>(p + i);
    if (acc == 0x7FFFFFFF) *sink = acc;
}

static constexpr size_t IFA_LDS_LIMIT = 160 * 1024;      // LDS per workgroup on gfx950 (MI355X_MICROARCH.md)
static long long *g_trace_ptr = nullptr;   // set by ifa_model_time_kernel when the "trace" option is on
static int num_cus() { return dec_num_cus(); }

// The fused GEMV of a weight tensor: int8-path formats stream their tiled copy (k_dec_gemv), everything else -- F16
// tensors, Q8_B32T1 / Q5_B32T1 / Q4_B16 / Q3_B32T1 / Q2_B32T1 -- the reference-layout bytes with fp16 activations
// (k_dec_gemv_h).  wbytes() hands out the matching pointer.
static bool fused_int8(int w_dtype) { return ax8_eligible(w_dtype); }
static const uint8_t *wbytes(const Tensor &t) { return (const uint8_t *)(fused_int8(t.dtype) ? t.tiled : t.data); }
static bool fused_ok(const Tensor &t, bool long_rows)
{
    if (!t.present()) return false;
    if (fused_int8(t.dtype)) return t.tiled && (long_rows ? dec_gemv_supported_long(t.dtype, t.cols) : dec_gemv_supported(t.dtype, t.cols));
    return dec_gemv_h_supported(t.dtype, t.cols) && (long_rows || t.cols <= 8192);
}

template <int EPI, int NORM>
static int launch_dec_gemv(int w_dtype, const DecGemvParams &P, int wgs_per_cu_opt, hipStream_t s)
{
    if (!fused_int8(w_dtype)) {
        if constexpr (NORM == 2 || epi_is_moe(EPI)) return ifa_fail(IFA_ERR_STATE, "fused GEMV: dtype %d has no kernel for this launch", w_dtype);
        else return dec_gemv_h_launch(w_dtype, EPI, NORM, P, s);
    }
    return dec_gemv_launch(w_dtype, EPI, NORM, P, wgs_per_cu_opt, s, g_trace_ptr);
}

static int lmhead_grid(const DecLmHeadParams &P, int wgs_per_cu_opt)
{
    const int nj = (P.cols / 8 + 63) / 64;
    const int R = nj <= 4 ? 2 : 1;
    const int nbatch = (P.rows + R - 1) / R;
    const int per_cu = wgs_per_cu_opt > 0 ? wgs_per_cu_opt : 2;
    return std::max(1, std::min(num_cus() * per_cu, (nbatch + DEC_WAVES - 1) / DEC_WAVES));
}

static int launch_lmhead(const DecLmHeadParams &P, int norm, int wgs_per_cu_opt, hipStream_t s, const DecStepTail *Z = nullptr)
{
    const int chunks = P.cols / 8;
    const int nj = (chunks + 63) / 64;
    if (P.cols % 8 != 0 || nj < 1 || nj > 16) return ifa_fail(IFA_ERR_ARG, "fused lm_head supports cols %% 8 == 0 and <= 8192 (got %d)", P.cols);
    const int R = nj <= 4 ? 2 : 1;
    const int nbatch = (P.rows + R - 1) / R;
    (void)nbatch;
    dim3 grid((unsigned)lmhead_grid(P, wgs_per_cu_opt));
    const size_t smem = (((size_t)P.cols * 2 + 15) & ~(size_t)15) + 132 * 4 + 16;
#define IFA_LM(NJV, RV) \
    case NJV: if (norm) k_dec_lmhead_f16<NJV, RV, 1><<<grid, dim3(DEC_THREADS), smem, s>>>(P); \
              else k_dec_lmhead_f16<NJV, RV, 0><<<grid, dim3(DEC_THREADS), smem, s>>>(P); break;
#define IFA_LMT(NJV, RV) \
    case NJV: if (norm) k_dec_lmhead_tail<NJV, RV, 1><<<grid, dim3(DEC_THREADS), smem, s>>>(P, *Z); \
              else k_dec_lmhead_tail<NJV, RV, 0><<<grid, dim3(DEC_THREADS), smem, s>>>(P, *Z); break;
    if (Z) {
        switch (nj) { IFA_LMT(1, 2) IFA_LMT(2, 2) IFA_LMT(3, 2) IFA_LMT(4, 2) IFA_LMT(5, 1) IFA_LMT(6, 1) IFA_LMT(7, 1) IFA_LMT(8, 1)
                      IFA_LMT(9, 1) IFA_LMT(10, 1) IFA_LMT(11, 1) IFA_LMT(12, 1) IFA_LMT(13, 1) IFA_LMT(14, 1) IFA_LMT(15, 1) IFA_LMT(16, 1) }
    } else
    switch (nj) { IFA_LM(1, 2) IFA_LM(2, 2) IFA_LM(3, 2) IFA_LM(4, 2) IFA_LM(5, 1) IFA_LM(6, 1) IFA_LM(7, 1) IFA_LM(8, 1)
                  IFA_LM(9, 1) IFA_LM(10, 1) IFA_LM(11, 1) IFA_LM(12, 1) IFA_LM(13, 1) IFA_LM(14, 1) IFA_LM(15, 1) IFA_LM(16, 1) }
#undef IFA_LM
#undef IFA_LMT
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// self_attn.post_norm / feed_forward.post_norm (OPT / BERT-style specs): the op-by-op layer only (layer_tail_ops)
static bool has_post_norms(const ifa_model *m)
{
    for (const Layer &L : m->layers) if (L.t[T_ATTN_POST_NORM].present() || L.t[T_FFN_POST_NORM].present()) return true;
    return false;
}

// Can the fused decode path run this model?  (otherwise decode falls back to forward())
static bool fused_supported(const ifa_model *m, std::string *why)
{
    const ifa_model_config &c = m->cfg;
    auto fail = [&](const char *s) { if (why) *why = s; return false; };
    if (has_post_norms(m)) return fail("post norms (self_attn.post_norm / feed_forward.post_norm) use the op-by-op path");
    if (c.experts > 64 || (c.experts > 0 && (c.moe_top_k < 1 || c.moe_top_k > 8))) return fail("MoE: experts / top_k out of range");
    if ((scale_on(c.attn_out_scale) || scale_on(c.ffn_out_scale) || scale_on(c.out_scale)) && c.tp_size > 1)
        return fail("output scales (attn_out_scale / ffn_out_scale / out_scale) on a partitioned model use the op-by-op path");
    if (c.experts > 0 && (c.norm_kind != 0 || c.parallel_attn || c.share_input)) return fail("MoE layers need the sequential RMS-norm wiring");
    if (!c.full_quant_gemv) return fail("full_quant_gemv disabled");
    if (c.head_dim != 32 && c.head_dim != 48 && c.head_dim != 64 && c.head_dim != 80 && c.head_dim != 96 && c.head_dim != 128)
        return fail("fused attention supports head_dim 32/48/64/80/96/128");
    if (c.kv_dtype == Q8_B32T2 && c.head_dim % 32 != 0) return fail("Q8 KV needs head_dim % 32 == 0");
    if (dec_attn_pv_smem(c.head_dim, c.max_ctx, DEC_ATTN_MAX_SPLITS) > IFA_LDS_LIMIT) return fail("max_context_len too large for the fused attention kernels' LDS (decode falls back to the op-by-op path)");
    if (c.dim % 32 != 0 || c.ffn % 32 != 0) return fail("dim/ffn must be multiples of 32");
    if (c.dim > 8192) return fail("fused norm prologue supports dim <= 8192");
    for (const Layer &L : m->layers) {
        const bool moe = c.experts > 0 && L.t[T_MOE_GATE].present();
        if (moe) {
            if ((int)L.experts.size() != c.experts * 3 || !L.moe_table) return fail("MoE: expert tensors missing");
            for (int e = 0; e < c.experts; e++)
                for (int k = 0; k < 3; k++) {
                    const Tensor &t = L.experts[(size_t)e * 3 + k];
                    if (!t.present() || !t.tiled || !(k == 1 ? dec_gemv_supported_long(t.dtype, t.cols) : dec_gemv_supported(t.dtype, t.cols)))
                        return fail("MoE: expert weights must be in an int8-GEMV format");
                    if (!same_fmt(t.dtype, L.experts[(size_t)(k == 1 ? 1 : 0)].dtype) || t.rows != L.experts[(size_t)k].rows) return fail("MoE: experts differ in dtype / shape");
                }
        }
        const int ids_dense[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W2};
        const int ids_moe[] = {T_WQ, T_WK, T_WV, T_WO};
        const int *ids = moe ? ids_moe : ids_dense;
        const int n_ids = moe ? 4 : 6;
        for (int ii = 0; ii < n_ids; ii++) {
            const int id = ids[ii];
            const Tensor &t = L.t[id];
            if (!t.present()) return fail("fused path: a layer's weight matrix is missing");
            const bool plain_input = id == T_WO || id == T_W2;      // neither normalised nor gated: long rows allowed
            if (!fused_ok(t, plain_input)) return fail("fused GEMV: columns out of range for this weight format (or cols % 8 != 0 for fp16 activations)");
        }
        if (L.t[T_W3].present() && (!fused_ok(L.t[T_W3], false) || !same_fmt(L.t[T_W3].dtype, L.t[T_W1].dtype))) return fail("w1/w3 dtype mismatch");
        if (!L.t[T_ATTN_NORM].present()) return fail("pre-norm weights required");
        if (!L.t[T_FFN_NORM].present() && !c.parallel_attn) return fail("ffn pre-norm weights required");
        // (wq / wk / wv of different formats -- grouped-query models under the tensor_quant_threshold rule -- get one launch each)
    }
    const Tensor &lm = m->g[T_LM_HEAD];
    // a pipeline stage (BY_LAYER partition) may hold neither embeddings nor lm_head: checked where they are used
    if (!lm.present()) { /* middle / first stage */ }
    else if (lm.dtype == F16) {
        if (lm.cols > 8192 || lm.cols % 8 != 0) return fail("fused F16 lm_head needs cols <= 8192");
    } else if (!fused_ok(lm, false) || !m->g[T_OUT_NORM].present()) {
        return fail("fused lm_head: columns out of range for its weight format (or no output norm)");
    }
    if (m->g[T_EMBD].present() && m->g[T_EMBD].dtype != F16) return fail("F16 embeddings required");
    return true;
}

// --------------------------------------------------- fused step (enqueue only)
// Std-norm models (Falcon, Bloom, OPT ...): the norm runs as the op-level kernel (same arithmetic as the op path by
// construction) into `dst`, and the GEMV that follows takes it without a norm prologue.
static int sep_norm(ifa_model *m, const half_t *x, const Tensor &w, const Tensor &b, half_t *dst)
{
    return ifa_layernorm(m->cfg.norm_kind, x, 1, (size_t)m->cfg.dim, w.data, b.data, 0.0f, m->cfg.eps, dst, (ifa_stream)m->stream);
}

static void attn_params(ifa_model *m, int l, DecAttnParams &A);
// Can layer l take the attention as the tail of its QKV launch?  (the one-workgroup-per-head attention, RMS-norm wiring, q / k / v
// of one int8-path format with a kernel instance, no tensor-parallel pending sum)
static bool qkv_attn_layer_ok(const ifa_model *m, int l, int *gk_out)
{
    const ifa_model_config &c = m->cfg;
    const Layer &L = m->layers[(size_t)l];
    if (c.norm_kind != 0 || c.tp_size > 1) return false;
    const Tensor &wq = L.t[T_WQ], &wk = L.t[T_WK], &wv = L.t[T_WV];
    if (!wq.present() || !wk.present() || !wv.present() || !wq.tiled || !wk.tiled || !wv.tiled) return false;
    if (!fused_int8(wq.dtype) || !same_fmt(wq.dtype, wk.dtype) || !same_fmt(wq.dtype, wv.dtype)) return false;
    if ((int)wq.rows != c.heads * c.head_dim || (int)wk.rows != c.kv_heads * c.head_dim || (int)wv.rows != c.kv_heads * c.head_dim) return false;
    int rw = 0;
    return dec_qkv_attn_supported(wq.dtype, (int)wq.cols, c.heads, c.kv_heads, c.head_dim, num_cus(), gk_out, &rw);
}

// decides qa_on for the next captured step and allocates what the fused launch needs (never under capture)
static int qkv_attn_ready(ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    int want = m->opt_fuse_attn && waits_enabled() && !m->attn_split && dec_attn_smem(c.head_dim, c.max_ctx, 256) <= IFA_LDS_LIMIT;
    int gk = 0;
    for (int l = 0; want && l < c.layers; l++) if (!qkv_attn_layer_ok(m, l, &gk)) want = 0;
    // a head's attention waits for the gk workgroups of its kv group: the grid (kv_heads * gk workgroups of 512 threads at 256
    // registers, one per CU) must be resident at once on the CUs this process may use (CU mask, partitioned device)
    if (want && (long long)c.kv_heads * gk > (long long)visible_cus()) want = 0;
    if (want && !m->qa_gran) {
        const size_t n = (size_t)c.layers * (size_t)(c.heads + 2 * c.kv_heads) * c.head_dim;
        IFA_HIP_CHECK(hipMalloc((void **)&m->qa_gran, n * 8));
        IFA_HIP_CHECK(hipMemsetAsync(m->qa_gran, 0, n * 8, m->stream));
        if (m->qa_call) { (void)hipFree(m->qa_call); m->qa_call = nullptr; }
        if (m->qa_err) { (void)hipFree(m->qa_err); m->qa_err = nullptr; }
        IFA_HIP_CHECK(hipMalloc((void **)&m->qa_call, 16));
        IFA_HIP_CHECK(hipMemsetAsync(m->qa_call, 0, 16, m->stream));
        IFA_HIP_CHECK(hipMalloc((void **)&m->qa_err, 16));
        IFA_HIP_CHECK(hipMemsetAsync(m->qa_err, 0, 16, m->stream));
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    }
    // the chained FFN launch: dense gated FFN behind an RMS pre-norm, sequential wiring, W1 / W3 / W2 (and Wo) of one format with an
    // instance, one workgroup per CU resident at once
    int want_ch = (m->opt_fuse_ffn && waits_enabled() && c.norm_kind == 0 && c.tp_size <= 1 && c.experts == 0
                   && !c.parallel_attn && !c.share_input && num_cus() <= visible_cus()) ? std::min(m->opt_fuse_ffn, 2) : 0;
    if (want_ch == 2 && (!m->attq || !m->opt_attn_q8 || c.head_dim % 32 != 0)) want_ch = 1;
    for (int l = 0; want_ch && l < c.layers; l++) {
        const Layer &L = m->layers[(size_t)l];
        const Tensor &wo = L.t[T_WO], &w1 = L.t[T_W1], &w3 = L.t[T_W3], &w2 = L.t[T_W2];
        if (!w1.present() || !w1.tiled || !w3.present() || !w3.tiled || !w2.present() || !w2.tiled || !L.t[T_FFN_NORM].present()
            || (int)w1.cols != c.dim || (int)w2.rows != c.dim || w2.cols != w1.rows || w3.rows != w1.rows || !same_fmt(w1.dtype, w3.dtype)
            || !dec_chain_supported(w1.dtype, w2.dtype, w1.dtype, c.dim, (int)w1.rows, false, c.dim, num_cus()))
            want_ch = 0;
        else if (want_ch == 2 && (!wo.present() || !wo.tiled || (int)wo.rows != c.dim
                                  || !dec_chain_supported(w1.dtype, w2.dtype, wo.dtype, c.dim, (int)w1.rows, true, (int)wo.cols, num_cus())))
            want_ch = 1;
    }
    if (want_ch && !m->ch_gran) {
        const size_t n = (size_t)c.layers * (size_t)(c.dim + (int)m->layers[0].t[T_W1].rows);
        IFA_HIP_CHECK(hipMalloc((void **)&m->ch_gran, n * 4));
        IFA_HIP_CHECK(hipMemsetAsync(m->ch_gran, 0, n * 4, m->stream));
        IFA_HIP_CHECK(hipMalloc((void **)&m->ch_flags, (size_t)c.layers * 2 * 1024 * 4));
        IFA_HIP_CHECK(hipMemsetAsync(m->ch_flags, 0, (size_t)c.layers * 2 * 1024 * 4, m->stream));
        if (!m->qa_call) { IFA_HIP_CHECK(hipMalloc((void **)&m->qa_call, 16)); IFA_HIP_CHECK(hipMemsetAsync(m->qa_call, 0, 16, m->stream)); }
        if (!m->qa_err) { IFA_HIP_CHECK(hipMalloc((void **)&m->qa_err, 16)); IFA_HIP_CHECK(hipMemsetAsync(m->qa_err, 0, 16, m->stream)); }
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    }
    if (want != m->qa_on || (want && gk != m->qa_gk) || want_ch != m->ch_on) {
        m->qa_on = want; m->qa_gk = gk; m->ch_on = want_ch; drop_graphs(m);
    }
    return IFA_OK;
}

static void qkv_params(ifa_model *m, int l, const half_t *x, DecGemvParams &P)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    memset(&P, 0, sizeof(P));
    P.x = x; P.norm_w = (const half_t *)L.t[T_ATTN_NORM].data; P.norm_b = (const half_t *)L.t[T_ATTN_NORM_B].data;
    P.multi_base = c.attn_norm_base; P.eps = c.eps; P.cols = c.dim; P.nblk = c.dim / 32;
    if (c.parallel_attn) P.xn_out = m->xn;
    const int ids[3] = {T_WQ, T_WK, T_WV}; const int bids[3] = {T_WQ_B, T_WK_B, T_WV_B};
    const size_t QDd = (size_t)c.heads * c.head_dim, KVDd = (size_t)c.kv_heads * c.head_dim;
    half_t *outs[3] = {m->dqkv, m->dqkv + QDd, m->dqkv + QDd + KVDd};
    for (int i = 0; i < 3; i++) {
        P.W0[i] = wbytes(L.t[ids[i]]); P.b0[i] = (const half_t *)L.t[bids[i]].data;
        P.y[i] = outs[i]; P.rows[i] = (int)L.t[ids[i]].rows;
    }
    P.nsets = 3;
}

// QKV GEMVs + the attention of every head in ONE launch (tag_add: distinct tags for the timing loop's repeated launches)
static int launch_qkv_attn(ifa_model *m, int l, const half_t *x, unsigned tag_add = 0)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; qkv_params(m, l, x, P);
    DecAttnParams A; attn_params(m, l, A);
    DecQkvAttnExtra E; memset(&E, 0, sizeof(E));
    E.gran = m->qa_gran + (size_t)l * (size_t)(c.heads + 2 * c.kv_heads) * c.head_dim;
    E.epoch = m->qa_call; E.epoch_add = tag_add; E.err = m->qa_err; E.timeout_us = m->opt_fuse_attn_timeout_us; E.gk = m->qa_gk;
    const int pb = (m->attn_pb == 64 || m->attn_pb == 128) ? m->attn_pb : 256;
    const bool kt = m->opt_attn_kt && !A.kv_q8 && dec_attn_smem(c.head_dim, c.max_ctx, pb) <= IFA_LDS_LIMIT;
    return dec_qkv_attn_launch(L.t[T_WQ].dtype, 1, A.kv_q8 != 0, pb, kt, P, A, E, c.max_ctx, m->stream);
}

static int launch_qkv(ifa_model *m, int l, const half_t *x)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; memset(&P, 0, sizeof(P));
    P.x = x; P.norm_w = (const half_t *)L.t[T_ATTN_NORM].data; P.norm_b = (const half_t *)L.t[T_ATTN_NORM_B].data;
    P.multi_base = c.attn_norm_base; P.eps = c.eps; P.cols = c.dim; P.nblk = c.dim / 32;
    if (c.parallel_attn) P.xn_out = m->xn;         // the normalised input: parallel-attention models feed it to the FFN
    const bool std_norm = c.norm_kind != 0;
    if (std_norm) {
        int rc = sep_norm(m, x, L.t[T_ATTN_NORM], L.t[T_ATTN_NORM_B], m->xn);
        if (rc) return rc;
        P.x = m->xn; P.norm_w = nullptr; P.norm_b = nullptr; P.xn_out = nullptr;
    }
    const int ids[3] = {T_WQ, T_WK, T_WV}; const int bids[3] = {T_WQ_B, T_WK_B, T_WV_B};
    const size_t QDd = (size_t)c.heads * c.head_dim, KVDd = (size_t)c.kv_heads * c.head_dim;
    half_t *outs[3] = {m->dqkv, m->dqkv + QDd, m->dqkv + QDd + KVDd};
    for (int i = 0; i < 3; i++) {
        P.W0[i] = wbytes(L.t[ids[i]]); P.b0[i] = (const half_t *)L.t[bids[i]].data;
        P.y[i] = outs[i]; P.rows[i] = (int)L.t[ids[i]].rows;
    }
    P.nsets = 3;
    if (m->pend.on && !std_norm) {      // x = pend.x + merged product: formed in this kernel's prologue, stored as the new layer input
        P.x = m->pend.x; P.x_add = m->pend.add; P.x_add_bias = m->pend.bias; P.xsum_out = m->pend.out;
        m->pend.on = false;
    }
    auto go = [&](int dtype, const DecGemvParams &Q) {
        if (std_norm) return launch_dec_gemv<EPI_PLAIN, 0>(dtype, Q, m->opt_rpw_qkv, m->stream);
        return launch_dec_gemv<EPI_PLAIN, 1>(dtype, Q, m->opt_rpw_qkv, m->stream);
    };
    if (same_fmt(L.t[T_WQ].dtype, L.t[T_WK].dtype) && same_fmt(L.t[T_WQ].dtype, L.t[T_WV].dtype)) return go(L.t[T_WQ].dtype, P);
    // mixed formats (e.g. wq quantised, wk / wv left F16 by the threshold rule): one launch per matrix; the first one forms
    // a pending sum, the others read the stored result
    for (int i = 0; i < 3; i++) {
        DecGemvParams Q = P;
        Q.nsets = 1; Q.W0[0] = P.W0[i]; Q.b0[0] = P.b0[i]; Q.y[0] = P.y[i]; Q.rows[0] = P.rows[i];
        if (i > 0) {
            Q.xn_out = nullptr;
            if (P.x_add) { Q.x = P.xsum_out; Q.x_add = nullptr; Q.x_add_bias = nullptr; Q.xsum_out = nullptr; }
        }
        int rc = go(L.t[ids[i]].dtype, Q);
        if (rc) return rc;
    }
    return IFA_OK;
}

static void attn_params(ifa_model *m, int l, DecAttnParams &A)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    const int rope_dims = (int)(c.head_dim * c.partial_rotary + 0.5f);
    memset(&A, 0, sizeof(A));
    A.q = m->dqkv; A.k_new = m->dqkv + (size_t)c.heads * c.head_dim; A.v_new = A.k_new + (size_t)c.kv_heads * c.head_dim; A.kcache = (uint8_t *)L.kcache; A.vcache = (uint8_t *)L.vcache;
    A.state = m->state; A.rope_tab = m->rope_tab; A.heads = c.heads; A.kv_heads = c.kv_heads;
    A.kv_q8 = c.kv_dtype == Q8_B32T2; A.kq_scale = c.use_alibi ? 1.0f : c.kq_scale;
    A.rope_order = c.rope_order; A.rope_cols = rope_dims;
    A.alibi = c.use_alibi; A.alibi_base = c.tp_rank * c.heads; A.alibi_total = c.heads * std::max(1, c.tp_size);
    A.out = m->att; A.max_ctx = c.max_ctx; A.xq = (c.head_dim % 32 == 0) ? m->attq : nullptr; A.trace = g_trace_ptr;
}

static int launch_attn(ifa_model *m, int l)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    const int rope_dims = (int)(c.head_dim * c.partial_rotary + 0.5f);
    DecAttnParams A; memset(&A, 0, sizeof(A));
    A.q = m->dqkv; A.k_new = m->dqkv + (size_t)c.heads * c.head_dim; A.v_new = A.k_new + (size_t)c.kv_heads * c.head_dim; A.kcache = (uint8_t *)L.kcache; A.vcache = (uint8_t *)L.vcache;
    A.state = m->state; A.rope_tab = m->rope_tab; A.heads = c.heads; A.kv_heads = c.kv_heads;
    A.kv_q8 = c.kv_dtype == Q8_B32T2; A.kq_scale = c.use_alibi ? 1.0f : c.kq_scale;
    A.rope_order = c.rope_order; A.rope_cols = rope_dims;
    A.alibi = c.use_alibi; A.alibi_base = c.tp_rank * c.heads; A.alibi_total = c.heads * std::max(1, c.tp_size);
    A.out = m->att; A.max_ctx = c.max_ctx; A.xq = (c.head_dim % 32 == 0) ? m->attq : nullptr; A.trace = g_trace_ptr;
    // the one-workgroup kernel keeps a head's score row [max_ctx] in LDS: past the device limit (160 KiB: ~75K tokens of
    // context at head_dim 128) the keys-split-over-workgroups kernels run from position 0 on (scores in global memory)
    const bool lds_split = dec_attn_smem(c.head_dim, c.max_ctx) > IFA_LDS_LIMIT;
    if (m->attn_split || lds_split) {
        // splits per head: 8, or what the decode call chose for the context it will reach (attn_split = 8 / 16 / 32); when only
        // the LDS forces the split (very large max_context_len) as many as keep a split's probabilities inside the LDS
        int nsp = m->attn_split > 1 ? m->attn_split : 8;
        while (nsp < DEC_ATTN_MAX_SPLITS && dec_attn_pv_smem(c.head_dim, c.max_ctx, nsp) > IFA_LDS_LIMIT) nsp *= 2;
        m->attn_ws.nsplits = nsp;
        const dim3 g2((unsigned)c.heads, (unsigned)nsp);
        const size_t psmem = dec_attn_pv_smem(c.head_dim, c.max_ctx, nsp);
        const size_t ssmem = dec_attn_scores_smem(c.head_dim, false);      // staging of 256 F16 key rows
        if (psmem > IFA_LDS_LIMIT) return ifa_fail(IFA_ERR_ARG, "fused attention: max_context_len %d needs %zu bytes of LDS per workgroup", c.max_ctx, psmem);
#define IFA_ATTN_S(HDV) \
    case HDV: if (A.kv_q8) { k_dec_attn_scores<HDV, true><<<g2, dim3(256), 16, m->stream>>>(A, m->attn_ws); \
                             if (psmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_pv<HDV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmem)); \
                             k_dec_attn_pv<HDV, true><<<g2, dim3(256), psmem, m->stream>>>(A, m->attn_ws); } \
              else { if (ssmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_scores<HDV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ssmem)); \
                     k_dec_attn_scores<HDV, false><<<g2, dim3(256), ssmem, m->stream>>>(A, m->attn_ws); \
                     if (psmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_pv<HDV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmem)); \
                     k_dec_attn_pv<HDV, false><<<g2, dim3(256), psmem, m->stream>>>(A, m->attn_ws); } \
              k_dec_attn_combine<HDV><<<dim3((unsigned)c.heads), dim3(HDV), 0, m->stream>>>(m->attn_ws, m->att, A.xq, c.heads); break;
        // head sizes that are not whole Q8 blocks (48, 80) exist with an F16 KV cache only
#define IFA_ATTN_SF(HDV) \
    case HDV: if (ssmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_scores<HDV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ssmem)); \
              k_dec_attn_scores<HDV, false><<<g2, dim3(256), ssmem, m->stream>>>(A, m->attn_ws); \
              if (psmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_pv<HDV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmem)); \
              k_dec_attn_pv<HDV, false><<<g2, dim3(256), psmem, m->stream>>>(A, m->attn_ws); \
              k_dec_attn_combine<HDV><<<dim3((unsigned)c.heads), dim3(HDV), 0, m->stream>>>(m->attn_ws, m->att, A.xq, c.heads); break;
        switch (c.head_dim) {
            IFA_ATTN_S(32) IFA_ATTN_S(64) IFA_ATTN_S(96) IFA_ATTN_S(128) IFA_ATTN_SF(48) IFA_ATTN_SF(80)
        default: return ifa_fail(IFA_ERR_ARG, "fused attention: head_dim %d", c.head_dim);
        }
#undef IFA_ATTN_SF
#undef IFA_ATTN_S
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    const int pb = (m->attn_pb == 64 || m->attn_pb == 128) ? m->attn_pb : 256;
    const bool kt = m->opt_attn_kt && !A.kv_q8 && (c.head_dim == 32 || c.head_dim == 64 || c.head_dim == 128)
        && dec_attn_smem(c.head_dim, c.max_ctx, pb) <= IFA_LDS_LIMIT;
    const size_t asmem = dec_attn_smem(c.head_dim, c.max_ctx, kt ? pb : 0);
    const dim3 grid((unsigned)c.heads), block(256);
    // (attention_lds_ok() routed contexts whose score row does not fit the 160 KiB LDS to the split kernels above)
#define IFA_ATTN_GO(HDV, Q8V, PBV, KTV) do { \
        auto kern = k_dec_attn<HDV, Q8V, false, PBV, KTV>; \
        if (asmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)asmem)); \
        kern<<<grid, block, asmem, m->stream>>>(A.q, A.kcache, A.vcache, A.heads, A.kv_heads, A); } while (0)
#define IFA_ATTN_PB(HDV, Q8V, KTV) do { if (pb == 64) IFA_ATTN_GO(HDV, Q8V, 64, KTV); else if (pb == 128) IFA_ATTN_GO(HDV, Q8V, 128, KTV); else IFA_ATTN_GO(HDV, Q8V, 256, KTV); } while (0)
    // head sizes with a power-of-two number of 16-byte pieces take the K rows through the LDS tile (F16 cache)
#define IFA_ATTN(HDV) \
    case HDV: if (A.kv_q8) IFA_ATTN_PB(HDV, true, false); else if (kt) IFA_ATTN_PB(HDV, false, true); else IFA_ATTN_PB(HDV, false, false); break;
#define IFA_ATTN_Q(HDV) \
    case HDV: if (A.kv_q8) IFA_ATTN_GO(HDV, true, 256, false); else IFA_ATTN_GO(HDV, false, 256, false); break;
#define IFA_ATTN_F(HDV) \
    case HDV: IFA_ATTN_GO(HDV, false, 256, false); break;
    switch (c.head_dim) {
        IFA_ATTN(32) IFA_ATTN(64) IFA_ATTN_Q(96) IFA_ATTN(128) IFA_ATTN_F(48) IFA_ATTN_F(80)
    default: return ifa_fail(IFA_ERR_ARG, "fused attention: head_dim %d", c.head_dim);
    }
#undef IFA_ATTN_Q
#undef IFA_ATTN_PB
#undef IFA_ATTN_GO
#undef IFA_ATTN_F
#undef IFA_ATTN
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// partial != nullptr (tensor parallel): write the un-merged product there, no bias, no residual
static int launch_wo(ifa_model *m, int l, const half_t *x, half_t *partial = nullptr)
{
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; memset(&P, 0, sizeof(P));
    P.x = m->att; P.cols = (int)L.t[T_WO].cols; P.nblk = P.cols / 32; P.eps = m->cfg.eps;
    P.W0[0] = wbytes(L.t[T_WO]); P.rows[0] = (int)L.t[T_WO].rows; P.nsets = 1;
    // the attention kernel left its output quantised (XqImage): the GEMV needs no prologue.  Rows longer than a lane's
    // register image (chunked kernel) keep the in-kernel quantiser
    const bool preq = m->attq && m->opt_attn_q8 && m->cfg.head_dim % 32 == 0 && P.cols == m->cfg.heads * m->cfg.head_dim && fused_int8(L.t[T_WO].dtype)
        && dec_gemv_supported(L.t[T_WO].dtype, (size_t)P.cols);
    if (preq) P.x = reinterpret_cast<const half_t *>(m->attq);      // NORM == 2 kernels read the quantised image through P.x
    if (partial) {
        P.y[0] = partial;
        return preq ? launch_dec_gemv<EPI_PLAIN, 2>(L.t[T_WO].dtype, P, m->opt_rpw_wo, m->stream)
                    : launch_dec_gemv<EPI_PLAIN, 0>(L.t[T_WO].dtype, P, m->opt_rpw_wo, m->stream);
    }
    P.b0[0] = (const half_t *)L.t[T_WO_B].data;
    P.y[0] = m->a; P.residual = x;
    if (scale_on(m->cfg.attn_out_scale)) P.pre_scale = m->cfg.attn_out_scale;      // Scale(self_att_out) fused in front of the residual add
    if (m->cfg.parallel_attn || m->cfg.share_input)      // the residual is added once, after the FFN (inference_worker.cc:847-851)
        return preq ? launch_dec_gemv<EPI_PLAIN, 2>(L.t[T_WO].dtype, P, m->opt_rpw_wo, m->stream)
                    : launch_dec_gemv<EPI_PLAIN, 0>(L.t[T_WO].dtype, P, m->opt_rpw_wo, m->stream);
    return preq ? launch_dec_gemv<EPI_RESIDUAL, 2>(L.t[T_WO].dtype, P, m->opt_rpw_wo, m->stream)
                : launch_dec_gemv<EPI_RESIDUAL, 0>(L.t[T_WO].dtype, P, m->opt_rpw_wo, m->stream);
}

static void moe_params(ifa_model *m, Layer &L, DecGemvParams &P, int slot, int tab_off)
{
    P.w_table = (const uint8_t *const *)L.moe_table;
    P.moe_sel = m->moe_route;
    P.moe_w = reinterpret_cast<const half_t *>(reinterpret_cast<const char *>(m->moe_route) + 32);
    P.moe_acc = m->f;
    P.moe_slot = slot; P.moe_tab_off = tab_off;
}

// moe_slot >= 0: the FFN of the expert the router put in that slot (weights through L.moe_table)
static int launch_ffn13(ifa_model *m, int l, int moe_slot = -1, const half_t *x_layer = nullptr, int moe_nslots = 1)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; memset(&P, 0, sizeof(P));
    P.x = m->a; P.norm_w = (const half_t *)L.t[T_FFN_NORM].data; P.norm_b = (const half_t *)L.t[T_FFN_NORM_B].data;
    P.eps = c.eps; P.cols = c.dim; P.nblk = c.dim / 32; P.act_kind = c.act_kind; P.multi_base = c.ffn_norm_base;
    if (moe_slot >= 0) {
        const Tensor &e1 = L.experts[0], &e3 = L.experts[2];
        moe_params(m, L, P, moe_slot, 0);
        P.W0[0] = (const uint8_t *)e1.tiled; P.W1 = (const uint8_t *)e3.tiled;   // (replaced by the table lookup)
        // moe_nslots router slots in one launch: set i = the expert of slot moe_slot + i, its product at t1 + i * ffn
        const int ns = std::max(1, std::min(3, moe_nslots));
        for (int i = 0; i < ns; i++) { P.y[i] = m->t1 + (size_t)i * e1.rows; P.rows[i] = (int)e1.rows; P.W0[i] = P.W0[0]; }
        P.nsets = ns;
        if (e3.present()) return launch_dec_gemv<EPI_MOE_GLU, 1>(e1.dtype, P, m->opt_rpw_ffn, m->stream);
        return launch_dec_gemv<EPI_MOE_ACT, 1>(e1.dtype, P, m->opt_rpw_ffn, m->stream);
    }
    P.W0[0] = wbytes(L.t[T_W1]); P.b0[0] = (const half_t *)L.t[T_W1_B].data;
    P.y[0] = m->t1; P.rows[0] = (int)L.t[T_W1].rows; P.nsets = 1;
    // FFN input (inference_worker.cc:853-872): the attention's normalised input (parallel attention), the layer input
    // (shared input) or the attention output + residual; then the FFN pre-norm if the model has one
    const half_t *ff_in = c.parallel_attn ? m->xn : (c.share_input ? x_layer : m->a);
    bool need_norm = L.t[T_FFN_NORM].present();
    P.x = ff_in;
    if (need_norm && c.norm_kind != 0) {
        int rc = sep_norm(m, ff_in, L.t[T_FFN_NORM], L.t[T_FFN_NORM_B], m->hn);
        if (rc) return rc;
        P.x = m->hn; need_norm = false;
    }
    if (!need_norm) { P.norm_w = nullptr; P.norm_b = nullptr; }
    const bool glu = L.t[T_W3].present();
    if (glu) { P.W1 = wbytes(L.t[T_W3]); P.b1 = (const half_t *)L.t[T_W3_B].data; }
    const int dtw = L.t[T_W1].dtype;
    if (need_norm && m->pend.on && P.x == m->pend.out) {     // the FFN input is the pending sum
        P.x = m->pend.x; P.x_add = m->pend.add; P.x_add_bias = m->pend.bias; P.xsum_out = m->pend.out;
        m->pend.on = false;
    }
    if (need_norm) return glu ? launch_dec_gemv<EPI_GLU, 1>(dtw, P, m->opt_rpw_ffn, m->stream) : launch_dec_gemv<EPI_ACT, 1>(dtw, P, m->opt_rpw_ffn, m->stream);
    return glu ? launch_dec_gemv<EPI_GLU, 0>(dtw, P, m->opt_rpw_ffn, m->stream) : launch_dec_gemv<EPI_ACT, 0>(dtw, P, m->opt_rpw_ffn, m->stream);
}

static int launch_w2(ifa_model *m, int l, half_t *xnext, half_t *partial = nullptr, int moe_slot = -1, bool moe_last = false,
                     const half_t *residual2 = nullptr, int moe_t1_slot = 0)
{
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; memset(&P, 0, sizeof(P));
    if (moe_slot >= 0) {
        const Tensor &e2 = L.experts[1];
        moe_params(m, L, P, moe_slot, 2);
        P.x = m->t1 + (size_t)moe_t1_slot * e2.cols; P.cols = (int)e2.cols; P.eps = m->cfg.eps;      // (the gated product of this slot)
        P.W0[0] = (const uint8_t *)e2.tiled; P.rows[0] = (int)e2.rows; P.nsets = 1;
        if (partial) {      // tensor parallel: accumulate the weighted shard products; merged and finished by the caller
            P.y[0] = partial; P.moe_acc = partial;
            return launch_dec_gemv<EPI_MOE_ACC, 0>(e2.dtype, P, m->opt_rpw_w2, m->stream);
        }
        P.y[0] = moe_last ? xnext : m->f; P.residual = m->a; P.residual2 = residual2;
        if (moe_last && scale_on(m->cfg.ffn_out_scale)) P.pre_scale = m->cfg.ffn_out_scale;
        if (moe_last && l + 1 == m->cfg.layers && scale_on(m->cfg.out_scale)) P.post_scale = m->cfg.out_scale;
        if (moe_last) return launch_dec_gemv<EPI_MOE_LAST, 0>(e2.dtype, P, m->opt_rpw_w2, m->stream);
        return launch_dec_gemv<EPI_MOE_ACC, 0>(e2.dtype, P, m->opt_rpw_w2, m->stream);
    }
    P.x = m->t1; P.cols = (int)L.t[T_W2].cols; P.nblk = P.cols / 32; P.eps = m->cfg.eps;
    P.W0[0] = wbytes(L.t[T_W2]); P.rows[0] = (int)L.t[T_W2].rows; P.nsets = 1;
    if (partial) {
        P.y[0] = partial;
        return launch_dec_gemv<EPI_PLAIN, 0>(L.t[T_W2].dtype, P, m->opt_rpw_w2, m->stream);
    }
    P.b0[0] = (const half_t *)L.t[T_W2_B].data;
    P.y[0] = xnext; P.residual = m->a; P.residual2 = residual2;     // + layer input for parallel / shared-input models
    if (scale_on(m->cfg.ffn_out_scale)) P.pre_scale = m->cfg.ffn_out_scale;                               // Scale(ff_out)
    if (l + 1 == m->cfg.layers && scale_on(m->cfg.out_scale)) P.post_scale = m->cfg.out_scale;        // Scale(last layer's output)
    return launch_dec_gemv<EPI_RESIDUAL, 0>(L.t[T_W2].dtype, P, m->opt_rpw_w2, m->stream);
}

// [Wo ->] W1 | W3 -> W2 of layer l as ONE launch (ifa_decode_chain.h); x = the layer input (Wo's residual), xnext = the layer output
static int launch_chain(ifa_model *m, int l, const half_t *x, half_t *xnext, unsigned tag_add = 0)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    const bool wo = m->ch_on == 2;
    DecGemvParams PW; memset(&PW, 0, sizeof(PW));       // launch_wo's EPI_RESIDUAL / NORM 2 parameters
    if (wo) {
        PW.x = reinterpret_cast<const half_t *>(m->attq); PW.cols = (int)L.t[T_WO].cols; PW.eps = c.eps;
        PW.W0[0] = wbytes(L.t[T_WO]); PW.rows[0] = (int)L.t[T_WO].rows;
        PW.b0[0] = (const half_t *)L.t[T_WO_B].data; PW.y[0] = m->a; PW.residual = x;
        if (scale_on(c.attn_out_scale)) PW.pre_scale = c.attn_out_scale;
    }
    DecGemvParams P; memset(&P, 0, sizeof(P));          // launch_ffn13's dense EPI_GLU, NORM 1 parameters
    P.x = m->a; P.norm_w = (const half_t *)L.t[T_FFN_NORM].data; P.norm_b = (const half_t *)L.t[T_FFN_NORM_B].data;
    P.eps = c.eps; P.cols = c.dim; P.act_kind = c.act_kind; P.multi_base = c.ffn_norm_base;
    P.W0[0] = wbytes(L.t[T_W1]); P.b0[0] = (const half_t *)L.t[T_W1_B].data; P.y[0] = m->t1; P.rows[0] = (int)L.t[T_W1].rows;
    P.W1 = wbytes(L.t[T_W3]); P.b1 = (const half_t *)L.t[T_W3_B].data;
    DecGemvParams Q; memset(&Q, 0, sizeof(Q));          // launch_w2's EPI_RESIDUAL parameters
    Q.x = m->t1; Q.cols = (int)L.t[T_W2].cols; Q.eps = c.eps;
    Q.W0[0] = wbytes(L.t[T_W2]); Q.rows[0] = (int)L.t[T_W2].rows; Q.b0[0] = (const half_t *)L.t[T_W2_B].data;
    Q.y[0] = xnext; Q.residual = m->a;
    if (scale_on(c.ffn_out_scale)) Q.pre_scale = c.ffn_out_scale;
    if (l + 1 == c.layers && scale_on(c.out_scale)) Q.post_scale = c.out_scale;
    DecChainExtra E; memset(&E, 0, sizeof(E));
    const size_t per = (size_t)c.dim + L.t[T_W1].rows;
    E.gran_a = m->ch_gran + (size_t)l * per; E.gran_h = E.gran_a + c.dim;
    E.flags_a = m->ch_flags + (size_t)l * 2048; E.flags_h = E.flags_a + 1024;
    E.state = m->state; E.epoch = m->qa_call; E.epoch_add = tag_add; E.err = m->qa_err; E.timeout_us = m->opt_fuse_attn_timeout_us;
    E.trace = g_trace_ptr; E.late_w2 = m->opt_chain_late_w2;
    return dec_chain_launch(L.t[T_W1].dtype, true, 1, wo, P, Q, wo ? &PW : nullptr, E, num_cus(), m->stream);
}

// can the step end in the one-launch tail?  (F16 lm_head behind the RMS / no final norm, the embedding table on this worker)
static bool step_tail_ok(const ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    return m->opt_step_tail && m->g[T_LM_HEAD].present() && m->g[T_LM_HEAD].dtype == F16 && m->g[T_EMBD].present()
        && !(c.norm_kind != 0 && m->g[T_OUT_NORM].present()) && c.dim % 8 == 0 && c.dim <= 8192;
}

static DecLmHeadParams lm_params(ifa_model *m, const half_t *x, half_t *logits_out)
{
    const ifa_model_config &c = m->cfg;
    DecLmHeadParams H; memset(&H, 0, sizeof(H));
    H.x = x; H.norm_w = (const half_t *)m->g[T_OUT_NORM].data; H.norm_b = (const half_t *)m->g[T_OUT_NORM_B].data;
    H.eps = c.eps; H.cols = c.dim; H.W = (const half_t *)m->g[T_LM_HEAD].data; H.logits = logits_out ? logits_out : m->logits;
    H.rows = (int)m->g[T_LM_HEAD].rows; H.xn_out = m->xn; H.multi_base = c.out_norm_base;
    return H;
}

// (allocates: not under capture)
static int step_tail_ready(ifa_model *m)
{
    const int want = step_tail_ok(m) ? 1 : 0;
    if (want != m->st_on) { m->st_on = want; drop_graphs(m); }
    if (!want) return IFA_OK;
    const int grid = lmhead_grid(lm_params(m, m->x, nullptr), m->opt_rpw_lm);
    if (grid > m->st_keys_n) {
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
        if (m->st_keys) (void)hipFree(m->st_keys);
        m->st_keys = nullptr; m->st_keys_n = 0;
        IFA_HIP_CHECK(hipMalloc((void **)&m->st_keys, sizeof(unsigned long long) * (size_t)grid));
        m->st_keys_n = grid;
    }
    if (!m->st_counter) {
        IFA_HIP_CHECK(hipMalloc((void **)&m->st_counter, 16));
        IFA_HIP_CHECK(hipMemsetAsync(m->st_counter, 0, 16, m->stream));
    }
    return IFA_OK;
}

// the last launch of a captured step: lm_head, argmax, state advance and the next step's gather (k_dec_lmhead_tail)
static int launch_lm_tail(ifa_model *m, const half_t *x)
{
    const ifa_model_config &c = m->cfg;
    const DecLmHeadParams H = lm_params(m, x, nullptr);
    DecStepTail Z; memset(&Z, 0, sizeof(Z));
    Z.state = m->state; Z.ring = ifa_model::RING; Z.keys = m->st_keys; Z.counter = m->st_counter;
    Z.embd = (const half_t *)m->g[T_EMBD].data; Z.vocab = (int)m->g[T_EMBD].rows; Z.x_out = m->x;
    Z.rope_tab = c.rope_order ? m->rope_tab : nullptr; Z.head_dim = c.head_dim; Z.theta = c.rope_theta;
    Z.rope_dims = (int)(c.head_dim * c.partial_rotary + 0.5f); Z.embd_scale = c.embd_scale;
    return launch_lmhead(H, m->g[T_OUT_NORM].present() ? 1 : 0, m->opt_rpw_lm, m->stream, &Z);
}

static int launch_gather(ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    // (debug_hidden_in: the layer input is what the caller stored in "x"; only the step's RoPE table is built)
    k_dec_gather<<<dim3(2), dim3(256), 0, m->stream>>>(m->opt_debug_hidden_in ? nullptr : (const half_t *)m->g[T_EMBD].data, m->state, c.dim, (int)m->g[T_EMBD].rows, m->x,
                                                       c.rope_order ? m->rope_tab : nullptr, c.head_dim, c.rope_theta,
                                                       (int)(c.head_dim * c.partial_rotary + 0.5f), c.embd_scale);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

static int launch_lm(ifa_model *m, const half_t *x, half_t *logits_out = nullptr)
{
    const ifa_model_config &c = m->cfg;
    const Tensor &lmt = m->g[T_LM_HEAD];
    if (lmt.dtype != F16) {      // quantised lm_head (<= 20-layer models, network_builder.cc:839-844): same fused GEMV as the layers
        DecGemvParams P; memset(&P, 0, sizeof(P));
        P.x = x; P.norm_w = (const half_t *)m->g[T_OUT_NORM].data; P.norm_b = (const half_t *)m->g[T_OUT_NORM_B].data;
        P.eps = c.eps; P.cols = c.dim; P.xn_out = m->xn; P.multi_base = c.out_norm_base;
        P.W0[0] = wbytes(lmt); P.rows[0] = (int)lmt.rows; P.nsets = 1;
        P.y[0] = logits_out ? logits_out : m->logits;
        return launch_dec_gemv<EPI_PLAIN, 1>(lmt.dtype, P, m->opt_rpw_lm, m->stream);
    }
    if (c.norm_kind != 0 && m->g[T_OUT_NORM].present()) {      // std final norm: op-level kernel, then the plain GEMV
        int rc = sep_norm(m, x, m->g[T_OUT_NORM], m->g[T_OUT_NORM_B], m->xn);
        if (rc) return rc;
        DecLmHeadParams H2; memset(&H2, 0, sizeof(H2));
        H2.x = m->xn; H2.eps = c.eps; H2.cols = c.dim; H2.W = (const half_t *)lmt.data; H2.logits = logits_out ? logits_out : m->logits;
        H2.rows = (int)lmt.rows;
        return launch_lmhead(H2, 0, m->opt_rpw_lm, m->stream);
    }
    return launch_lmhead(lm_params(m, x, logits_out), m->g[T_OUT_NORM].present() ? 1 : 0, m->opt_rpw_lm, m->stream);
}

// Device-side counterpart of the host routing in moe_ffn (HostTensorOpr::BuildRowsForMoE, host_tensor_opr.cc:190-244):
// top-k by repeated first-maximum, probabilities below 1e-5 dropped, optional renormalisation, experts then visited in
// ascending id order.  Unused slots get weight 0 (hfma(y, 0, acc) == acc).  One thread: E <= 64, k <= 8.
// top-k of one row by ONE wave, lane e = expert e with probability p (lanes >= E: -inf): k_moe_topk's rules -- repeated first
// maximum, probabilities below 1e-5 dropped, optional renormalisation in pick order, kept experts in ascending id, unused
// slots expert 0 / weight 0.  (The one-thread form walked local arrays that live in scratch: ~20 us of dependent loads.)
__device__ __forceinline__ void moe_topk_wave(float p, int lane, int E, int top_k, int norm_topk, int *__restrict__ sel, half_t *__restrict__ wout, int unused_id = 0)
{
    bool used = lane >= E;
    int idx[8]; float w[8];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { idx[k] = 0x7FFFFFFF; w[k] = 0.0f; }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= top_k || k >= E) break;
        const float mx = wave_max(used ? -INFINITY : p);
        const unsigned long long cand = __ballot(!used && p == mx);
        if (!cand) break;
        const int best = __ffsll((long long)cand) - 1;
        if (lane == best) used = true;
        const float pb = __shfl(p, best);
        if (pb < 0.00001f) continue;
#pragma unroll
        for (int j = 0; j < 8; j++) if (j == n) { idx[j] = best; w[j] = pb; }
        n++;
    }
    if (norm_topk && n > 0) {
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) if (j < n) sum = sum + w[j];
#pragma unroll
        for (int j = 0; j < 8; j++) if (j < n) w[j] = w[j] / sum;
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j >= n) continue;
            int rank = 0;
#pragma unroll
            for (int j2 = 0; j2 < 8; j2++) rank += (j2 < n && idx[j2] < idx[j]) ? 1 : 0;
            sel[rank] = idx[j]; wout[rank] = f2h(w[j]);
        }
        for (int slot = n; slot < top_k; slot++) { sel[slot] = unused_id; wout[slot] = (half_t)0; }
    }
}

__global__ void __launch_bounds__(64) k_moe_topk(const half_t *__restrict__ probs_h, int E, int top_k, int norm, int *__restrict__ sel, half_t *__restrict__ wout)
{
    const int lane = threadIdx.x & 63;      // launched with one wave
    moe_topk_wave(lane < E ? h2f(probs_h[lane]) : -INFINITY, lane, E, top_k, norm, sel, wout);
}

// The router of a fused decode step in ONE launch (one workgroup of 8 waves): RMS norm of the layer's FFN input, the F16
// gate GEMV, softmax, top-k -- each with the arithmetic of the kernel it replaces (k_layernorm<0>: canonical RMS order;
// k_gemv_f16w: lane l takes chunks l, l + 64, ... as one fp32 fma chain, then the wave butterfly; k_softmax: 32-lane
// max / sum trees, half-rounded exponentials; k_moe_topk), so the routing is bit-identical to the four-launch sequence.
// The normalised input is also written out (hn) for parity checks.  cols % 8 == 0, cols <= 16384, E <= 64.
__global__ void __launch_bounds__(512) k_dec_moe_router(const half_t *__restrict__ x, const half_t *__restrict__ nw, const half_t *__restrict__ nb,
                                                        float multi_base, float eps, int cols, const half_t *__restrict__ gate_w, int E, int top_k,
                                                        int norm_topk, half_t *__restrict__ hn_out, half_t *__restrict__ probs_out,
                                                        int *__restrict__ sel, half_t *__restrict__ wout, int unused_id)
{
    // one workgroup per row (the batched step: blockIdx.x = query; a decode step: one row); rows are `cols` apart, a row's
    // routing top_k slots apart, unused slots carry `unused_id` (0 for the fused decode step: weight 0 makes them no-ops; -1 for
    // the list builder of the batched step, k_moe_route_rows' convention)
    x += (size_t)blockIdx.x * cols;
    if (hn_out) hn_out += (size_t)blockIdx.x * cols;
    if (probs_out) probs_out += (size_t)blockIdx.x * E;
    sel += (size_t)blockIdx.x * top_k; wout += (size_t)blockIdx.x * top_k;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *xs = reinterpret_cast<half_t *>(smem);                                                    // [cols]
    float *part = reinterpret_cast<float *>(smem + (((size_t)cols * 2 + 15) & ~(size_t)15));         // [64] group sums
    half_t *probs = reinterpret_cast<half_t *>(part + 64);                                            // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunks = cols >> 3;
    // the first gate row of every wave does not depend on the input: requested now (clamped, unconditional)
    u32x4 w_first[8];
    {
        const u32x4 *wrow = reinterpret_cast<const u32x4 *>(gate_w + (size_t)min(wave, E - 1) * cols);
#pragma unroll
        for (int j = 0; j < 8; j++) w_first[j] = wrow[min(lane + 64 * j, chunks - 1)];
    }
    // ---- RMS norm (or a plain copy when the layer has no FFN norm: nw == nullptr and eps < 0)
    const bool do_norm = eps >= 0.0f;
    for (int k = 0; k * 512 < chunks; k++) {
        const int c = tid + k * 512;
        rms_h8 v8;
#pragma unroll
        for (int e = 0; e < 8; e++) v8[e] = (half_t)0;
        if (c < chunks) { v8 = *reinterpret_cast<const rms_h8 *>(x + (size_t)c * 8); *reinterpret_cast<rms_h8 *>(xs + (size_t)c * 8) = v8; }
        const float pg = wave_sum(rms_chunk_sq(v8));
        if (lane == 0) part[wave + k * 8] = pg;
    }
    __syncthreads();
    if (do_norm) {
        const float scale = rms_scale_of(rms_total(part, (chunks + 63) >> 6), cols, eps);
        for (int c = tid; c < chunks; c += 512) {
            const rms_h8 v8 = *reinterpret_cast<const rms_h8 *>(xs + (size_t)c * 8);
            rms_h8 w8 = v8, b8 = v8;           // (one 16-byte request each: element-wise 2-byte loads behind branches took ~12 us)
            if (nw) w8 = *reinterpret_cast<const rms_h8 *>(nw + (size_t)c * 8);
            if (nb) b8 = *reinterpret_cast<const rms_h8 *>(nb + (size_t)c * 8);
            rms_h8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const half_t we = w8[e], be = b8[e];
                o[e] = f2h(rms_apply((float)v8[e], scale, nw ? &we : nullptr, nb ? &be : nullptr, multi_base));
            }
            *reinterpret_cast<rms_h8 *>(xs + (size_t)c * 8) = o;
            if (hn_out) *reinterpret_cast<rms_h8 *>(hn_out + (size_t)c * 8) = o;
        }
        __syncthreads();
    }
    // ---- gate GEMV: a wave per expert row, eight 16-byte requests in flight per lane (wave w's first expert row was
    // requested at the top of the kernel, before the norm)
    for (int e = wave; e < E; e += 8) {
        const u32x4 *wrow = reinterpret_cast<const u32x4 *>(gate_w + (size_t)e * cols);
        const u32x4 *xv = reinterpret_cast<const u32x4 *>(xs);
        float acc = 0.0f;
        for (int c0 = lane; c0 < chunks; c0 += 512) {
            u32x4 wr[8];
            if (e == wave && c0 == lane) {
#pragma unroll
                for (int j = 0; j < 8; j++) wr[j] = w_first[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) wr[j] = wrow[min(c0 + 64 * j, chunks - 1)];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) if (c0 + 64 * j < chunks) acc = dot8_f16(wr[j], xv[c0 + 64 * j], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) probs[e] = f2h(acc);
    }
    __syncthreads();
    // ---- softmax over the E gate values (k_softmax with one row, scale 1, no mask) and the top-k, by the first 32 lanes
    if (tid < 32) {
        float mx = -INFINITY;
        for (int xi = tid; xi < E; xi += 32) mx = fmaxf(mx, 1.0f * h2f(probs[xi]));
#pragma unroll
        for (int mk = 16; mk > 0; mk >>= 1) mx = fmaxf(mx, __shfl_xor(mx, mk, 32));
        float sum = 0.0f;
        for (int xi = tid; xi < E; xi += 32) {
            const float v = 1.0f * h2f(probs[xi]);
            const float ex = expf(v - mx);
            sum = sum + ex;
            probs[xi] = f2h(ex);
        }
#pragma unroll
        for (int mk = 16; mk > 0; mk >>= 1) sum = sum + __shfl_xor(sum, mk, 32);
        const float inv = 1.0f / sum;
        for (int xi = tid; xi < E; xi += 32) { const half_t pr = f2h(h2f(probs[xi]) * inv); probs[xi] = pr; if (probs_out) probs_out[xi] = pr; }
    }
    __syncthreads();
    // ---- top-k by wave 0, lane e = expert e (k_moe_topk's rules: repeated first maximum, probabilities below 1e-5 dropped,
    // optional renormalisation in pick order, kept experts in ascending id, unused slots expert 0 / weight 0).  The
    // one-thread form walks local arrays that live in scratch: ~20 us of dependent scratch loads per layer.
    if (wave == 0) moe_topk_wave(lane < E ? h2f(probs[lane]) : -INFINITY, lane, E, top_k, norm_topk, sel, wout, unused_id);
}

extern "C" int ifa_add_layernorm(int kind, const void *a, const void *addend, size_t rows, size_t cols, const void *w, const void *b,
                                 float multi_base, float eps, void *sum_out, void *y, ifa_stream stream);
extern "C" int ifa_rope_qk_store(void *q, void *k, const void *v, int head_dim, int heads, int kv_heads, int tokens, int pos0, float theta,
                                 int order, float partial_rotary_factor, void *kcache_rows, void *vcache_rows, size_t cache_row_elems,
                                 ifa_stream stream);
extern "C" int ifa_activation_mul(int kind, const void *a, const void *b, size_t n, void *c, ifa_stream stream);
extern "C" int ifa_argmax_rows(const void *logits, size_t n, size_t row_stride, size_t rows, int *out_dev, const int *excluded_dev, ifa_stream stream);
extern "C" int ifa_gemm_rows_q4(const void *Wt_tiled, size_t rows, size_t cols, const void *x_f16, size_t tokens,
                                const void *bias_f16, void *y_f16, ifa_stream stream);
namespace ifa {     // ifa_moe.hip / ifa_gemm.hip / ifa_gemv.hip
int moe_build_lists(const int *sel, const void *wsel, int T, int top_k, int E, int tile_rows, int small_max, int *idx, void *wdev, int *epos,
                    MoeTile *tiles, MoeSingle *singles, MoeTile *smalls, int *counts, hipStream_t s);
int gemm_rows_q4_grouped_cap(size_t cols);
bool gemm_rows_use_mfma();
bool gemm_rows_mfma_ok(size_t rows, size_t cols, size_t tokens);
int gemm_rows_mfma_grouped(const MoeSmallGroup &grp, size_t rows, size_t cols, const void *X, void *Y, int max_groups, int max_rows, hipStream_t s);
int gemm_rows_mo_grouped(const MoeSmallGroup &grp, size_t rows, size_t cols, const void *X, void *Y, int max_groups, int glu, int act_kind, hipStream_t s);
int gemm_rows_q4_grouped(const MoeSmallGroup &grp, size_t rows, size_t cols, const void *X, void *Y, int max_groups, hipStream_t s);
int moe_gather(const void *src, const int *idx, const int *counts, int max_entries, int dim, void *dst, hipStream_t s);
int moe_combine(const void *y, const int *epos, const void *wsel, int T, int top_k, int dim, void *out, hipStream_t s, const void *residual = nullptr);
int gemm_q_grouped(int w_dtype, const MoeGroup &grp, size_t N, size_t K, const void *X, void *Y, int max_tiles, int tile_rows, hipStream_t s);
int gemv_ax8_grouped(int w_dtype, const MoeGroup &grp, size_t rows, size_t cols, const void *xq8_rows, void *y_rows, int max_singles,
                     hipStream_t s);
}
static int matmul(ifa_model *m, const half_t *A, int T, const Tensor &W, const Tensor &bias, half_t *C);
static int norm_rows(ifa_model *m, const half_t *x, int T, const Tensor &w, const Tensor &b, half_t *y, float base = 0.0f);

// router of one MoE layer on the device: the same norm / GEMV / softmax kernels the op path runs, then k_moe_topk
static int launch_moe_router(ifa_model *m, int l)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    int rc;
    Tensor none;
    const Tensor &gw = L.t[T_MOE_GATE];
    if (m->opt_moe_router_fused && c.norm_kind == 0 && gw.dtype == F16 && c.dim % 8 == 0 && c.dim <= 16384 && c.experts <= 64 && (int)gw.cols == c.dim) {
        const bool has_norm = L.t[T_FFN_NORM].present();
        const size_t smem = (((size_t)c.dim * 2 + 15) & ~(size_t)15) + 64 * 4 + 64 * 2;
        k_dec_moe_router<<<1, 512, smem, m->stream>>>(m->a, has_norm ? (const half_t *)L.t[T_FFN_NORM].data : nullptr,
                                                      has_norm ? (const half_t *)L.t[T_FFN_NORM_B].data : nullptr, c.ffn_norm_base, has_norm ? c.eps : -1.0f,
                                                      c.dim, (const half_t *)gw.data, c.experts, c.moe_top_k, c.moe_norm_topk, m->hn, m->moe_gate, m->moe_route,
                                                      reinterpret_cast<half_t *>(reinterpret_cast<char *>(m->moe_route) + 32), 0);
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    const half_t *ff_n = m->a;
    if (L.t[T_FFN_NORM].present()) {
        if ((rc = norm_rows(m, m->a, 1, L.t[T_FFN_NORM], L.t[T_FFN_NORM_B], m->hn, c.ffn_norm_base))) return rc;
        ff_n = m->hn;
    }
    if ((rc = matmul(m, ff_n, 1, L.t[T_MOE_GATE], none, m->moe_gate))) return rc;
    if ((rc = ifa_softmax(m->moe_gate, c.experts, 1, 1, -1, 1.0f, (ifa_stream)m->stream))) return rc;
    k_moe_topk<<<1, 64, 0, m->stream>>>(m->moe_gate, c.experts, c.moe_top_k, c.moe_norm_topk, m->moe_route,
                                        reinterpret_cast<half_t *>(reinterpret_cast<char *>(m->moe_route) + 32));
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

static int enqueue_fused_step(ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    hipStream_t s = m->stream;
    int rc;
    // st_on: the previous step's last launch (or ifa_model_decode, for a call's first step) has gathered this step's input
    if (!m->st_on && (rc = launch_gather(m))) return rc;
    half_t *x = m->x, *xnext = m->x2;
    const int l_first = std::min(std::max(m->opt_debug_layer0, 0), c.layers - 1);
    const int n_layers = (m->opt_debug_layers > 0 && l_first + m->opt_debug_layers < c.layers) ? l_first + m->opt_debug_layers : c.layers;
    for (int l = l_first; l < n_layers; l++) {
        if (m->qa_on) {
            if ((rc = launch_qkv_attn(m, l, x))) return rc;
        } else {
            if ((rc = launch_qkv(m, l, x))) return rc;
            if ((rc = launch_attn(m, l))) return rc;
        }
        if (m->ch_on) {      // [Wo ->] W1 | W3 -> W2 as one launch
            if (m->ch_on == 1 && (rc = launch_wo(m, l, x))) return rc;
            if ((rc = launch_chain(m, l, x, xnext))) return rc;
            std::swap(x, xnext);
            continue;
        }
        if ((rc = launch_wo(m, l, x))) return rc;
        Layer &L = m->layers[(size_t)l];
        if (c.experts > 0 && L.t[T_MOE_GATE].present()) {
            if ((rc = launch_moe_router(m, l))) return rc;
            // the gated products of up to three router slots share one launch, then one W2 launch per slot (each accumulates
            // hfma(product, w, acc) in slot order)
            for (int k0 = 0; k0 < c.moe_top_k; k0 += 3) {
                const int ns = std::min(3, c.moe_top_k - k0);
                if ((rc = launch_ffn13(m, l, k0, nullptr, ns))) return rc;
                for (int k = k0; k < k0 + ns; k++)
                    if ((rc = launch_w2(m, l, xnext, nullptr, k, k + 1 == c.moe_top_k, nullptr, k - k0))) return rc;
            }
        } else {
            const bool extra = c.parallel_attn || c.share_input;
            if ((rc = launch_ffn13(m, l, -1, x))) return rc;
            if ((rc = launch_w2(m, l, xnext, nullptr, -1, false, extra ? x : nullptr))) return rc;
        }
        std::swap(x, xnext);
    }
    if (m->st_on) return launch_lm_tail(m, x);
    if ((rc = launch_lm(m, x))) return rc;
    k_dec_argmax_advance<<<dim3(1), dim3(1024), 0, s>>>(m->logits, (int)m->g[T_LM_HEAD].rows, m->state, ifa_model::RING);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// ------------------------------------------------ op-by-op forward (any T)
static int ensure_scratch(ifa_model *m, int T)
{
    if (T <= m->scratch_tokens) return IFA_OK;
    const ifa_model_config &c = m->cfg;
    auto re = [&](half_t *&p, size_t n) -> int {
        if (p) IFA_HIP_CHECK(hipFree(p));
        p = nullptr;
        IFA_HIP_CHECK(hipMalloc((void **)&p, n * sizeof(half_t)));
        return IFA_OK;
    };
    const size_t D = c.dim, QD = (size_t)c.heads * c.head_dim, KVD = (size_t)c.kv_heads * c.head_dim, F = c.ffn;
    size_t maxcols = std::max(std::max(D, QD), F);
    int rc;
    if ((rc = re(m->x, T * D)) || (rc = re(m->x2, T * D)) || (rc = re(m->xn, T * D)) || (rc = re(m->hn, T * D)) || (rc = re(m->pn, T * D))
        || (rc = re(m->q, T * QD)) || (rc = re(m->k, T * KVD)) || (rc = re(m->v, T * KVD)) || (rc = re(m->att, T * QD))
        || (rc = re(m->a, T * D)) || (rc = re(m->f, T * D)) || (rc = re(m->t1, std::max<size_t>(T, 3) * F)) || (rc = re(m->t2, T * F))
        || (rc = re(m->logits, (size_t)T * c.vocab)))
        return rc;
    if (!m->dqkv) IFA_HIP_CHECK(hipMalloc((void **)&m->dqkv, (QD + 2 * KVD) * sizeof(half_t)));
    if (T <= 32 && (size_t)T > m->bqkv_rows) {
        if (m->bqkv) IFA_HIP_CHECK(hipFree(m->bqkv));
        if (m->brope) IFA_HIP_CHECK(hipFree(m->brope));
        IFA_HIP_CHECK(hipMalloc((void **)&m->bqkv, 32 * (QD + 2 * KVD) * sizeof(half_t)));
        IFA_HIP_CHECK(hipMalloc((void **)&m->brope, 32 * (size_t)c.head_dim * sizeof(float)));
        m->bqkv_rows = 32;
    }
    if (c.experts > 0) {
        const size_t cap = (size_t)T * (size_t)std::max(1, c.moe_top_k);
        if ((rc = re(m->moe_gate, (size_t)T * c.experts)) || (rc = re(m->moe_out, (size_t)T * D)) || (rc = re(m->moe_in, (size_t)T * D))
            || (rc = re(m->moe_wdev, cap)))
            return rc;
        if (m->moe_idx) IFA_HIP_CHECK(hipFree(m->moe_idx));
        IFA_HIP_CHECK(hipMalloc((void **)&m->moe_idx, cap * sizeof(int)));
        if (m->moe_pin) IFA_HIP_CHECK(hipHostFree(m->moe_pin));
        IFA_HIP_CHECK(hipHostMalloc((void **)&m->moe_pin, cap * (sizeof(int) + 2) + 16, hipHostMallocDefault));
        if (!m->moe_route) { IFA_HIP_CHECK(hipMalloc((void **)&m->moe_route, 64)); IFA_HIP_CHECK(hipMemsetAsync(m->moe_route, 0, 64, m->stream)); }
        // device-routed path: every expert's rows at once (cap entries)
        void **raw[] = {(void **)&m->moe_sel, (void **)&m->moe_epos, (void **)&m->moe_counts, (void **)&m->moe_xq_in, (void **)&m->moe_xq_mid,
                        &m->moe_tiles, &m->moe_singles, &m->moe_smalls};
        for (void **p : raw) if (*p) { IFA_HIP_CHECK(hipFree(*p)); *p = nullptr; }
        IFA_HIP_CHECK(hipMalloc((void **)&m->moe_sel, cap * sizeof(int)));
        IFA_HIP_CHECK(hipMalloc((void **)&m->moe_epos, cap * sizeof(int)));
        IFA_HIP_CHECK(hipMalloc((void **)&m->moe_counts, 16 * sizeof(int)));
        IFA_HIP_CHECK(hipMalloc((void **)&m->moe_xq_in, cap * (D / 32 + 1) * 34));
        IFA_HIP_CHECK(hipMalloc((void **)&m->moe_xq_mid, cap * (F / 32 + 1) * 34));
        IFA_HIP_CHECK(hipMalloc(&m->moe_tiles, (cap / 64 + (size_t)c.experts + 1) * sizeof(MoeTile)));
        IFA_HIP_CHECK(hipMalloc(&m->moe_singles, ((size_t)c.experts + 1) * sizeof(MoeSingle)));
        IFA_HIP_CHECK(hipMalloc(&m->moe_smalls, ((size_t)c.experts + 1) * sizeof(MoeTile)));
        if ((rc = re(m->moe_selw, cap)) || (rc = re(m->moe_g1, cap * F)) || (rc = re(m->moe_g3, cap * F)) || (rc = re(m->moe_gin, cap * D))
            || (rc = re(m->moe_gout, cap * D)))
            return rc;
    }
    if (m->xq) IFA_HIP_CHECK(hipFree(m->xq));
    IFA_HIP_CHECK(hipMalloc((void **)&m->xq, (maxcols / 32 + 1) * 34));
    if (!m->attq) IFA_HIP_CHECK(hipMalloc((void **)&m->attq, xq_image_bytes(c.heads * c.head_dim)));
    if (m->tokens_dev) IFA_HIP_CHECK(hipFree(m->tokens_dev));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tokens_dev, sizeof(int) * (size_t)T));
    m->scratch_tokens = T;
    // buffers moved: any captured graph is stale
    drop_graphs(m);
    return IFA_OK;
}

__global__ void __launch_bounds__(256) k_gather_rows(const half_t *__restrict__ embd, const int *__restrict__ tokens,
                                                     int T, int dim, int vocab, half_t *__restrict__ x, float embd_scale)
{
    const int t = blockIdx.y;
    int tok = tokens[t];
    tok = min(max(tok, 0), vocab - 1);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < dim; c += gridDim.x * blockDim.x) {
        const half_t e = embd[(size_t)tok * dim + c];
        x[(size_t)t * dim + c] = embd_scale != 0.0f ? f2h(h2f(e) * embd_scale) : e;       // LinearNorm (inference_worker.cc:447-451)
    }
}

// MatrixMultiplicationEx + MatrixMultiplication dispatch (inference_worker.cc:2337-2432)
static int matmul(ifa_model *m, const half_t *A, int T, const Tensor &W, const Tensor &bias, half_t *C)
{
    if (!W.present()) return ifa_fail(IFA_ERR_STATE, "missing weight tensor");
    const void *b = bias.present() ? bias.data : nullptr;
    const size_t K = W.cols, N = W.rows;
    ifa_stream s = m->stream;
    const bool use_gemv = (T == 1) && (K % 32 == 0);
    if (use_gemv && W.dtype != F16 && m->cfg.full_quant_gemv && ax8_eligible(W.dtype)) {
        int rc = ifa_quantize_act_q8(A, 1, K, m->xq, s);
        if (rc) return rc;
        return ifa_gemv(W.dtype, W.data, N, K, Q8_B32T2, m->xq, b, C, s);
    }
    // a handful of rows (dynamic batching, very short prompts): weight-streaming kernel on the tiled layout
    if (T >= 2 && T <= 16 && is_q4(W.dtype) && W.tiled && m->opt_gemm_rows) {
        int rc = ifa_gemm_rows_q4(W.tiled, N, K, A, (size_t)T, b, C, s);
        if (rc != IFA_ERR_STATE) return rc;
    }
    // T > 1: MFMA GEMM with the dequantisation fused in (the reference dequantises the whole
    // tensor and calls cublasGemmEx; same arithmetic: half weights x half activations, fp32 accumulate)
    if (T > 1 && K % 8 == 0) return ifa_gemm(W.dtype, W.data, N, K, A, (size_t)T, b, C, s);
    // T == 1 with ineligible types: weights dequantised to half, fp32 accumulate per row
    for (int t = 0; t < T; t++) {
        int rc = ifa_gemv(W.dtype, W.data, N, K, F16, A + (size_t)t * K, b, C + (size_t)t * N, s);
        if (rc) return rc;
    }
    return IFA_OK;
}

static int norm_rows(ifa_model *m, const half_t *x, int T, const Tensor &w, const Tensor &b, half_t *y, float base)
{
    return ifa_layernorm(m->cfg.norm_kind, x, (size_t)T, (size_t)m->cfg.dim, w.present() ? w.data : nullptr,
                         b.present() ? b.data : nullptr, base, m->cfg.eps, y, m->stream);
}

// w2 . (act(w1 . x) [* (w3 . x)])   (ProcessGpuLayer_FeedForward, inference_worker.cc:1726-1922)
// ---- distributed greedy argmax over a vocabulary-sharded lm_head (one workgroup per row)
// (value, global id) of the best allowed logit of this rank's shard; first maximum wins
__global__ void __launch_bounds__(1024) k_tp_local_best(const half_t *__restrict__ v_all, size_t row_stride, int n, int vocab_offset,
                                                        const int *__restrict__ excl, float *__restrict__ best_all)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    const half_t *v = v_all + (size_t)blockIdx.x * row_stride;
    float *best_out = best_all + 2 * (size_t)blockIdx.x;
    const int ne = excl ? min(max(excl[0], 0), 3) : 0;
    const int e0 = ne > 0 ? excl[1] : -1, e1 = ne > 1 ? excl[2] : -1, e2 = ne > 2 ? excl[3] : -1;
    float best = -INFINITY; int besti = 0x7FFFFFFF;
    argmax_scan(v, (size_t)n, e0, e1, e2, (int)threadIdx.x, (int)blockDim.x, best, besti, vocab_offset);
#pragma unroll
    for (int mk = 32; mk > 0; mk >>= 1) {
        const float ob = __shfl_xor(best, mk); const int oi = __shfl_xor(besti, mk);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        best_out[0] = best;
        reinterpret_cast<int *>(best_out)[1] = besti;
    }
}

// the group's choice per row: highest value, lowest id among equals.  gathered: [rank][row][2]
__global__ void k_tp_pick(const float *__restrict__ gathered, int nranks, int n_rows, int *__restrict__ token)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    float best = -INFINITY; int besti = 0x7FFFFFFF;
    for (int k = 0; k < nranks; k++) {
        const float f = gathered[2 * ((size_t)k * n_rows + r)];
        const int gid = reinterpret_cast<const int *>(gathered)[2 * ((size_t)k * n_rows + r) + 1];
        if (f > best || (f == best && gid < besti)) { best = f; besti = gid; }
    }
    token[r] = besti == 0x7FFFFFFF ? 0 : besti;
}

// scratch of the distributed argmax for n_rows rows
static int tp_argmax_scratch(ifa_model *m, size_t n_rows)
{
    if (n_rows <= m->tp_rows_cap) return IFA_OK;
    if (m->tp_best) IFA_HIP_CHECK(hipFree(m->tp_best));
    if (m->tp_gather) IFA_HIP_CHECK(hipFree(m->tp_gather));
    if (m->tp_tok) IFA_HIP_CHECK(hipFree(m->tp_tok));
    m->tp_best = nullptr; m->tp_gather = nullptr; m->tp_tok = nullptr;
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_best, 8 * n_rows));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_gather, 8 * 64 * n_rows));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_tok, 4 * n_rows));
    m->tp_rows_cap = n_rows;
    drop_graphs(m);
    return IFA_OK;
}

// tokens[r] (device, m->tp_tok) = the group's greedy choice for row r of this rank's logits shard [n_rows][row_stride]
static int tp_pick_rows(ifa_model *m, const ifa_tp_topology &t, const half_t *shard, size_t row_stride, int shard_rows, int n_rows)
{
    int rc = tp_argmax_scratch(m, (size_t)n_rows);
    if (rc) return rc;
    ifa_stream s = (ifa_stream)m->stream;
    const int tp_size = t.tp ? ifa_comm_size(t.tp) : 1;
    const bool merge = t.tp && (tp_size > 1 || t.force_collectives);
    k_tp_local_best<<<dim3((unsigned)n_rows), 1024, 0, m->stream>>>(shard, row_stride, shard_rows, t.vocab_offset, m->state + 3, m->tp_best);
    IFA_LAUNCH_CHECK();
    const float *gathered = m->tp_best;
    int n_g = 1;
    if (merge) {
        if ((rc = ifa_allgather(t.tp, m->tp_best, m->tp_gather, 8 * (size_t)n_rows, s))) return rc;
        gathered = m->tp_gather; n_g = tp_size;
    }
    k_tp_pick<<<dim3((unsigned)((n_rows + 63) / 64)), 64, 0, m->stream>>>(gathered, n_g, n_rows, m->tp_tok);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// ---- tensor-parallel T > 1 / batched steps: the same op sequence, with the reference's merge
// (DistributeAndMergeTensors, inference_worker.cc:2148-2195) after the two column-sliced products of a layer
static bool tp_merging(const ifa_model *m)
{
    const ifa_tp_topology *t = m->topo;
    return t && t->tp && (ifa_comm_size(t->tp) > 1 || t->force_collectives);
}
// buf[T][dim] holds this rank's partial product (computed WITHOUT bias): sum over the group, then the bias once
static int tp_merge_rows(ifa_model *m, half_t *buf, int T, const Tensor &bias)
{
    if (!tp_merging(m)) return IFA_OK;
    const size_t D = (size_t)m->cfg.dim;
    int rc = ifa_allreduce_sum_f16(m->topo->tp, buf, buf, (size_t)T * D, (ifa_stream)m->stream);
    if (rc) return rc;
    if (bias.present()) return ifa_add(buf, bias.data, (size_t)T * D, D, buf, (ifa_stream)m->stream);
    return IFA_OK;
}

static int ffn_dense(ifa_model *m, const half_t *x, int T, const Tensor &w1, const Tensor &b1, const Tensor &w3, const Tensor &b3,
                     const Tensor &w2, const Tensor &b2, half_t *out)
{
    int rc;
    ifa_stream s = (ifa_stream)m->stream;
    if ((rc = matmul(m, x, T, w1, b1, m->t1))) return rc;
    if (w3.present()) {
        if ((rc = matmul(m, x, T, w3, b3, m->t2))) return rc;
        if ((rc = ifa_activation_mul(m->cfg.act_kind, m->t1, m->t2, (size_t)T * w1.rows, m->t1, s))) return rc;
    } else if ((rc = ifa_activation(m->cfg.act_kind, 0, m->t1, (size_t)T, w1.rows, m->t1, s))) return rc;
    return matmul(m, m->t1, T, w2, b2, out);
}

// Mixture of experts (ProcessGpuLayer_Moe, inference_worker.cc:1924-2146): router GEMV -> softmax -> D2H ->
// host top-k (HostTensorOpr::BuildRowsForMoE, host_tensor_opr.cc:190-244: probabilities below 1e-5 are dropped,
// optional renormalisation) -> the selected experts' FFNs in ascending expert order, each row on the T=1
// path -> B[row] = hfma(out, weight, B[row]) (AddByRowIdx_Kernel).  Result in m->f.
static bool moe_device_ok(const ifa_model *m, const Layer &L);
static int moe_ffn_device(ifa_model *m, Layer &L, const half_t *ff_n, int T, const half_t *pre_norm = nullptr, const half_t *residual = nullptr, half_t *out = nullptr);

static int moe_ffn(ifa_model *m, Layer &L, const half_t *ff_n, int T)
{
    if (T > 1 && moe_device_ok(m, L)) return moe_ffn_device(m, L, ff_n, T);
    const ifa_model_config &c = m->cfg;
    const size_t D = (size_t)c.dim; const int E = c.experts;
    ifa_stream s = (ifa_stream)m->stream;
    int rc;
    IFA_REQUIRE(E <= 64 && c.moe_top_k >= 1 && c.moe_top_k <= 8, "MoE: experts %d / top_k %d out of range", E, c.moe_top_k);
    IFA_REQUIRE((int)L.experts.size() == E * 3, "MoE: expert tensors missing");
    Tensor none;
    half_t *gate = m->moe_gate;
    if ((rc = matmul(m, ff_n, T, L.t[T_MOE_GATE], none, gate))) return rc;
    if ((rc = ifa_softmax(gate, E, T, 1, -1, 1.0f, s))) return rc;
    std::vector<uint16_t> probs_h((size_t)T * E);
    IFA_HIP_CHECK(hipMemcpyAsync(probs_h.data(), gate, probs_h.size() * 2, hipMemcpyDeviceToHost, m->stream));
    IFA_HIP_CHECK(hipMemsetAsync(m->f, 0, (size_t)T * D * 2, m->stream));
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    // per expert: the rows routed to it (token order) and their weights  (BuildRowsForMoE)
    std::vector<std::vector<int>> rows((size_t)E);
    std::vector<std::vector<uint16_t>> wts((size_t)E);
    for (int t = 0; t < T; t++) {
        float probs[64]; int idx[8]; float w[8]; bool used[64] = {false};
        for (int e = 0; e < E; e++) probs[e] = (float)__builtin_bit_cast(_Float16, probs_h[(size_t)t * E + e]);
        int n = 0;
        for (int k = 0; k < c.moe_top_k && k < E; k++) {          // first maximum wins, like the host sort
            int best = -1;
            for (int e = 0; e < E; e++) if (!used[e] && (best < 0 || probs[e] > probs[best])) best = e;
            if (best < 0) break;
            used[best] = true;
            if (probs[best] < 0.00001f) continue;
            idx[n] = best; w[n] = probs[best]; n++;
        }
        if (c.moe_norm_topk && n > 0) {
            float sum = 0.0f;
            for (int i = 0; i < n; i++) sum = sum + w[i];
            for (int i = 0; i < n; i++) w[i] = w[i] / sum;
        }
        for (int j = 0; j < n; j++) {
            const _Float16 wh = (_Float16)w[j];
            rows[(size_t)idx[j]].push_back(t);
            wts[(size_t)idx[j]].push_back(__builtin_bit_cast(uint16_t, wh));
        }
    }
    // one upload of every (row, weight) list, experts back to back
    const size_t cap = (size_t)T * (size_t)c.moe_top_k;
    int *pin_rows = m->moe_pin; uint16_t *pin_w = reinterpret_cast<uint16_t *>(m->moe_pin + cap);
    size_t off = 0;
    std::vector<size_t> start((size_t)E, 0);
    for (int e = 0; e < E; e++) {
        start[(size_t)e] = off;
        for (size_t r = 0; r < rows[(size_t)e].size(); r++) { pin_rows[off + r] = rows[(size_t)e][r]; pin_w[off + r] = wts[(size_t)e][r]; }
        off += rows[(size_t)e].size();
    }
    IFA_HIP_CHECK(hipMemcpyAsync(m->moe_idx, pin_rows, off * sizeof(int), hipMemcpyHostToDevice, m->stream));
    IFA_HIP_CHECK(hipMemcpyAsync(m->moe_wdev, pin_w, off * 2, hipMemcpyHostToDevice, m->stream));
    // expert by expert (ascending id): gather its rows, FFN on them as one matrix (T = 1 -> GEMV path, else the
    // MFMA GEMM, exactly the split MatrixMultiplication makes), scatter-add weight * output
    for (int e = 0; e < E; e++) {
        const int n = (int)rows[(size_t)e].size();
        if (n == 0) continue;
        const int *idx_dev = m->moe_idx + start[(size_t)e];
        k_gather_rows<<<dim3(4, (unsigned)n), dim3(256), 0, m->stream>>>(ff_n, idx_dev, n, (int)D, T, m->moe_in, 0.0f);
        IFA_LAUNCH_CHECK();
        const Tensor *ew = &L.experts[(size_t)e * 3];
        if ((rc = ffn_dense(m, m->moe_in, n, ew[0], none, ew[2], none, ew[1], none, m->moe_out))) return rc;
        if ((rc = ifa_add_by_row_index(m->f, m->moe_out, (size_t)n, D, idx_dev, m->moe_wdev + start[(size_t)e], s))) return rc;
    }
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));      // the pinned lists are reused by the next MoE layer
    return IFA_OK;
}

// The same layer without the host (T > 1 rows; ifa_moe.h): routing and the per-expert row lists are built on the device,
// the rows of ALL experts are gathered once (experts ascending, token order inside an expert -- the reference's order), and
// each of the three products is ONE grouped launch over the experts with >= 2 rows (MFMA GEMM tiles, the reference's T > 1
// branch: F16 activations on dequantised weights) plus ONE over the single-row experts (the int8-activation GEMV of its
// T = 1 branch, bit-identical to the op-level kernel).  No D2H copy, no stream synchronisation.  Result in m->f.
static bool moe_device_ok(const ifa_model *m, const Layer &L)
{
    const ifa_model_config &c = m->cfg;
    if (!m->opt_moe_device || !L.moe_table_aos || (int)L.experts.size() != c.experts * 3) return false;
    const Tensor &w1 = L.experts[0], &w2 = L.experts[1], &w3 = L.experts[2];
    if (!w3.present() || !ax8_eligible(w1.dtype) || !m->cfg.full_quant_gemv || w1.cols % 32 || w2.cols % 32) return false;
    for (int e = 0; e < c.experts; e++)
        for (int k3 = 0; k3 < 3; k3++) {
            const Tensor &t = L.experts[(size_t)e * 3 + k3], &r = L.experts[(size_t)k3];
            if (!t.present() || t.dtype != r.dtype || t.rows != r.rows || t.cols != r.cols) return false;
        }
    return true;
}

static int max_smalls_possible(bool rows_kernel, int E, int cap) { return rows_kernel ? std::min(E, cap / 2) : 0; }
// (also called by the batched step before it starts a capture: nothing is created inside one)
static int ensure_side_stream(ifa_model *m)
{
    if (m->side_stream) return IFA_OK;
    IFA_HIP_CHECK(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
    IFA_HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    IFA_HIP_CHECK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    return IFA_OK;
}

// can the rows of a batched step be routed by ONE launch (k_dec_moe_router, a workgroup per row: norm, F16 gate GEMV, softmax, top-k)?
static bool moe_router_rows_ok(const ifa_model *m, const Layer &L, int T)
{
    const ifa_model_config &c = m->cfg;
    const Tensor &gw = L.t[T_MOE_GATE];
    return m->opt_moe_router_fused && T <= 32 && c.norm_kind == 0 && gw.dtype == F16 && c.dim % 8 == 0 && c.dim <= 16384 && c.experts <= 64
        && (int)gw.cols == c.dim && L.t[T_FFN_NORM].present();
}

// pre_norm: the rows BEFORE the FFN norm (ff_n is then where the normalised rows go): the batched step's router launch does the norm too
// residual / out: the layer's residual Add in the combine launch, result in `out` (default: the FFN output alone in m->f)
static int moe_ffn_device(ifa_model *m, Layer &L, const half_t *ff_n, int T, const half_t *pre_norm, const half_t *residual, half_t *out)
{
    const ifa_model_config &c = m->cfg;
    const size_t D = (size_t)c.dim, F = L.experts[0].rows;
    const int E = c.experts, K = c.moe_top_k, cap = T * K;
    ifa_stream s = (ifa_stream)m->stream;
    int rc;
    Tensor none;
    if (pre_norm) {
        // norm + gate + softmax + top-k of every row as one launch instead of four (each with the arithmetic of the decode step's
        // router: the gate product is the F16 GEMV's fp32 chain, not the GEMM tile's): 25 -> 7 us per layer at 8 queries
        const size_t smem = (((size_t)c.dim * 2 + 15) & ~(size_t)15) + 64 * 4 + 64 * 2;
        k_dec_moe_router<<<dim3((unsigned)T), 512, smem, m->stream>>>(pre_norm, (const half_t *)L.t[T_FFN_NORM].data, (const half_t *)L.t[T_FFN_NORM_B].data,
                                                                      c.ffn_norm_base, c.eps, c.dim, (const half_t *)L.t[T_MOE_GATE].data, E, K, c.moe_norm_topk,
                                                                      const_cast<half_t *>(ff_n), m->moe_gate, m->moe_sel, (half_t *)m->moe_selw, -1);
        IFA_LAUNCH_CHECK();
    } else {
        if ((rc = matmul(m, ff_n, T, L.t[T_MOE_GATE], none, m->moe_gate))) return rc;
        if ((rc = ifa_softmax(m->moe_gate, E, T, 1, -1, 1.0f, s))) return rc;
        if ((rc = ifa_moe_route_topk(m->moe_gate, (size_t)T, E, K, c.moe_norm_topk, m->moe_sel, m->moe_selw, s))) return rc;
    }
    // rows per expert on average >= 96: 128-row tiles (each decoded weight block feeds four MFMA tiles); else 64-row split-K tiles
    const int tile_rows = (cap / std::max(1, E) >= 96) ? 128 : 64;
    // a handful of rows per expert (dynamic batching): experts with 2..small_max rows stream their tiled Q4 weights once
    // (ifa_gemm_rows.hip) instead of filling a 64-row MFMA tile with mostly padding
    const int wdt = L.experts[0].dtype;
    const bool rows_mfma = gemm_rows_use_mfma() && gemm_rows_mfma_ok(F, D, 2) && gemm_rows_mfma_ok(D, F, 2);     // matrix-core variant: up to 16 rows
    const bool rows_kernel = is_q4(wdt) && m->opt_gemm_rows && L.moe_table && cap <= 8 * E
        && (rows_mfma || (gemm_rows_q4_grouped_cap(D) > 0 && gemm_rows_q4_grouped_cap(F) > 0));
    const int small_max = rows_kernel ? 8 : 0;
    auto rows_grouped = [&](const MoeSmallGroup &q, size_t rows, size_t cols, const void *X, void *Y, int ng) {
        return rows_mfma ? gemm_rows_mfma_grouped(q, rows, cols, X, Y, ng, small_max, m->stream) : gemm_rows_q4_grouped(q, rows, cols, X, Y, ng, m->stream);
    };
    if ((rc = moe_build_lists(m->moe_sel, m->moe_selw, T, K, E, tile_rows, small_max, m->moe_idx, m->moe_wdev, m->moe_epos, (MoeTile *)m->moe_tiles,
                              (MoeSingle *)m->moe_singles, (MoeTile *)m->moe_smalls, m->moe_counts, m->stream))) return rc;
    MoeSmallGroup sg;
    sg.smalls = (const MoeTile *)m->moe_smalls; sg.counts = m->moe_counts; sg.wtab_tiled = (const uint8_t *const *)L.moe_table; sg.which_tiled = 0;
    // MO copies of the experts (ensure_mo, built by the batched step before its capture): the small groups then take ONE launch
    // for w1 / w3 with the gated product as its output, and one for w2 -- instead of three launches and an element-wise pass
    sg.wtab_mo = (rows_mfma && m->opt_rows_mo) ? (const uint8_t *const *)L.moe_table_mo : nullptr;
    const bool smalls_mo = sg.wtab_mo != nullptr && max_smalls_possible(rows_kernel, E, cap) > 0;
    const int max_smalls = rows_kernel ? std::min(E, cap / 2) : 0;
    if ((rc = moe_gather(ff_n, m->moe_idx, m->moe_counts, cap, (int)D, m->moe_gin, m->stream))) return rc;
    MoeGroup g;
    g.tiles = (const MoeTile *)m->moe_tiles; g.singles = (const MoeSingle *)m->moe_singles; g.counts = m->moe_counts;
    g.wtab = (const uint8_t *const *)L.moe_table_aos; g.on = 1;
    // (T <= small_max: no expert can collect more rows than the small groups take -- the tile list is empty, its launches are skipped)
    const int max_tiles = (small_max > 0 && T <= small_max) ? 0 : cap / tile_rows + E, max_singles = std::min(E, cap);
    // Single-row experts (round 4): when no expert can collect a tile of rows (max_tiles == 0: a batched decode step) and the small
    // groups gate their own products (MO copies), the singles are the only rows the quantiser / element-wise launches below serve --
    // they then take two launches of the decode GEMV's structure on the tiled expert tables (ifa_decode_singles.h) instead of six
    if (m->opt_moe_singles && max_tiles == 0 && (smalls_mo || max_smalls == 0) && L.moe_table && dec_singles_supported(wdt, F, D, true)
        && dec_singles_supported(wdt, D, F, false)) {
        DecSinglesParams S; memset(&S, 0, sizeof(S));
        S.singles = (const MoeSingle *)m->moe_singles; S.counts = m->moe_counts; S.wtab = (const uint8_t *const *)L.moe_table; S.act_kind = c.act_kind;
        // the singles' two launches on the side stream, the small groups' two on the main one: disjoint rows of g1 / gout, joined in
        // front of the combine (inside a capture the fork / join become graph edges)
        hipStream_t ss = m->stream;
        if (m->opt_moe_overlap && max_smalls) {
            if ((rc = ensure_side_stream(m))) return rc;
            ss = m->side_stream;
            IFA_HIP_CHECK(hipEventRecord(m->ev_fork, m->stream));
            IFA_HIP_CHECK(hipStreamWaitEvent(ss, m->ev_fork, 0));
        }
        S.which = 0; S.X = m->moe_gin; S.ldx = (int)D; S.Y = m->moe_g1; S.ldy = (int)F; S.rows = (int)F; S.cols = (int)D; S.nblk = (int)(D / 32);
        if ((rc = dec_singles_launch(wdt, S, true, max_singles, ss))) return rc;                      // act(w1 x) * (w3 x)
        S.which = 2; S.X = m->moe_g1; S.ldx = (int)F; S.Y = m->moe_gout; S.ldy = (int)D; S.rows = (int)D; S.cols = (int)F; S.nblk = (int)(F / 32);
        if ((rc = dec_singles_launch(wdt, S, false, max_singles, ss))) return rc;                     // w2
        if (max_smalls) {
            sg.which_tiled = 0;
            if ((rc = gemm_rows_mo_grouped(sg, F, D, m->moe_gin, m->moe_g1, max_smalls, 1, c.act_kind, m->stream))) return rc;
            sg.which_tiled = 2;
            if ((rc = gemm_rows_mo_grouped(sg, D, F, m->moe_g1, m->moe_gout, max_smalls, 0, c.act_kind, m->stream))) return rc;
        }
        if (ss != m->stream) {
            IFA_HIP_CHECK(hipEventRecord(m->ev_join, ss));
            IFA_HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_join, 0));
        }
        return moe_combine(m->moe_gout, m->moe_epos, m->moe_selw, T, K, (int)D, out ? out : m->f, m->stream, residual);
    }
    // single-row experts take the quantised row (TensorOpr::Quantize in front of Gemv_AX, inference_worker.cc:1772-1774)
    if ((rc = ifa_quantize_act_q8(m->moe_gin, (size_t)cap, D, m->moe_xq_in, s))) return rc;
    g.which = 0;
    if ((rc = gemm_q_grouped(wdt, g, F, D, m->moe_gin, m->moe_g1, max_tiles, tile_rows, m->stream))) return rc;
    if ((rc = gemv_ax8_grouped(wdt, g, F, D, m->moe_xq_in, m->moe_g1, max_singles, m->stream))) return rc;
    if (max_smalls && !smalls_mo && (rc = rows_grouped(sg, F, D, m->moe_gin, m->moe_g1, max_smalls))) return rc;
    g.which = 2; sg.which_tiled = 1;
    if ((rc = gemm_q_grouped(wdt, g, F, D, m->moe_gin, m->moe_g3, max_tiles, tile_rows, m->stream))) return rc;
    if ((rc = gemv_ax8_grouped(wdt, g, F, D, m->moe_xq_in, m->moe_g3, max_singles, m->stream))) return rc;
    if (max_smalls && !smalls_mo && (rc = rows_grouped(sg, F, D, m->moe_gin, m->moe_g3, max_smalls))) return rc;
    if ((rc = ifa_activation_mul(c.act_kind, m->moe_g1, m->moe_g3, (size_t)cap * F, m->moe_g1, s))) return rc;
    if (max_smalls && smalls_mo) {      // (after the element-wise pass over all rows: the small groups' rows of g1 are written here, gated)
        sg.which_tiled = 0;
        if ((rc = gemm_rows_mo_grouped(sg, F, D, m->moe_gin, m->moe_g1, max_smalls, 1, c.act_kind, m->stream))) return rc;
    }
    if ((rc = ifa_quantize_act_q8(m->moe_g1, (size_t)cap, F, m->moe_xq_mid, s))) return rc;
    g.which = 1;
    if ((rc = gemm_q_grouped(wdt, g, D, F, m->moe_g1, m->moe_gout, max_tiles, tile_rows, m->stream))) return rc;
    if ((rc = gemv_ax8_grouped(wdt, g, D, F, m->moe_xq_mid, m->moe_gout, max_singles, m->stream))) return rc;
    sg.which_tiled = 2;
    if (max_smalls && smalls_mo) { if ((rc = gemm_rows_mo_grouped(sg, D, F, m->moe_g1, m->moe_gout, max_smalls, 0, c.act_kind, m->stream))) return rc; }
    else if (max_smalls && (rc = rows_grouped(sg, D, F, m->moe_g1, m->moe_gout, max_smalls))) return rc;
    return moe_combine(m->moe_gout, m->moe_epos, m->moe_selw, T, K, (int)D, out ? out : m->f, m->stream, residual);
}

// Everything of a layer behind the attention product in m->a (bias added / shards merged): TensorOpr::Scale of the attention
// output, the residual wiring, the optional post norms, the FFN (dense or mixture of experts) and the adds in front of what
// follows -- ProcessGpuLayer, inference_worker.cc:841-965.  Shared by the prompt path and the batched step.  x: the layer
// input (on return: the layer output, m->x / m->f exchanged); attn_in: the attention's normalised input (parallel attention feeds
// it to the FFN); xn_ready: m->xn holds the next norm's output already (fused into the last Add).
//   self_attn.post_norm (:857-866): residual = a (+ x); a' = Norm(residual); the FFN reads a'; is_attn_post_as_residual picks a' as
//   the residual too.  feed_forward.post_norm (:954-965): the layer output is Norm(ffn out + residual [+ x]).
static int layer_tail_ops(ifa_model *m, int l, int T, half_t *&x, const half_t *attn_in, bool &xn_ready)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    ifa_stream s = m->stream;
    const size_t D = c.dim;
    const Tensor none;
    const bool merging = tp_merging(m), seq_wiring = !c.parallel_attn && !c.share_input;
    const bool a_post = L.t[T_ATTN_POST_NORM].present(), f_post = L.t[T_FFN_POST_NORM].present();
    int rc;
    if (scale_on(c.attn_out_scale) && (rc = ifa_scale(m->a, c.attn_out_scale, (size_t)T * D, m->a, s))) return rc;
    const half_t *ff_in = c.parallel_attn ? attn_in : (c.share_input ? x : m->a);
    const half_t *ff_n = ff_in;
    const half_t *residual = m->a;
    if (a_post) {
        if (seq_wiring && (rc = ifa_add(x, m->a, (size_t)T * D, 0, m->a, s))) return rc;
        if ((rc = norm_rows(m, m->a, T, L.t[T_ATTN_POST_NORM], L.t[T_ATTN_POST_NORM_B], m->pn, 0.0f))) return rc;
        if (m->opt_attn_post_as_residual) residual = m->pn;
        if (!c.parallel_attn && !c.share_input) ff_in = m->pn;
        ff_n = ff_in;
        if (L.t[T_FFN_NORM].present()) {
            if ((rc = norm_rows(m, ff_in, T, L.t[T_FFN_NORM], L.t[T_FFN_NORM_B], m->hn, c.ffn_norm_base))) return rc;
            ff_n = m->hn;
        }
    } else if (seq_wiring && L.t[T_FFN_NORM].present()) {          // Add(x, attn out) + ffn norm
        if ((rc = ifa_add_layernorm(c.norm_kind, x, m->a, (size_t)T, D, L.t[T_FFN_NORM].data, L.t[T_FFN_NORM_B].present() ? L.t[T_FFN_NORM_B].data : nullptr,
                                    c.ffn_norm_base, c.eps, m->a, m->hn, s))) return rc;
        ff_n = m->hn;
    } else {
        if (seq_wiring && (rc = ifa_add(x, m->a, (size_t)T * D, 0, m->a, s))) return rc;
        if (L.t[T_FFN_NORM].present()) {
            if ((rc = norm_rows(m, ff_in, T, L.t[T_FFN_NORM], L.t[T_FFN_NORM_B], m->hn, c.ffn_norm_base))) return rc;
            ff_n = m->hn;
        }
    }
    if (c.experts > 0 && L.t[T_MOE_GATE].present()) {
        if ((rc = moe_ffn(m, L, ff_n, T))) return rc;
        if ((rc = tp_merge_rows(m, m->f, T, none))) return rc;          // every expert sliced like the dense FFN: one merge of the weighted sums
    } else {
        if ((rc = ffn_dense(m, ff_n, T, L.t[T_W1], L.t[T_W1_B], L.t[T_W3], L.t[T_W3_B], L.t[T_W2], merging ? none : L.t[T_W2_B], m->f))) return rc;
        if ((rc = tp_merge_rows(m, m->f, T, L.t[T_W2_B]))) return rc;
    }
    if (scale_on(c.ffn_out_scale) && (rc = ifa_scale(m->f, c.ffn_out_scale, (size_t)T * D, m->f, s))) return rc;
    // Add(ffn out, residual) + the norm in front of what comes next: the next layer's attention norm, or the output norm
    const bool last_layer = l + 1 == c.layers;
    const Tensor &nw = last_layer ? m->g[T_OUT_NORM] : m->layers[(size_t)l + 1].t[T_ATTN_NORM];
    const Tensor &nb = last_layer ? m->g[T_OUT_NORM_B] : m->layers[(size_t)l + 1].t[T_ATTN_NORM_B];
    xn_ready = false;
    if (!a_post && !f_post && seq_wiring && nw.present() && !(last_layer && scale_on(c.out_scale))) {
        if ((rc = ifa_add_layernorm(c.norm_kind, m->f, m->a, (size_t)T, D, nw.data, nb.present() ? nb.data : nullptr,
                                    last_layer ? c.out_norm_base : c.attn_norm_base, c.eps, m->f, m->xn, s))) return rc;
        xn_ready = true;
    } else {
        if ((rc = ifa_add(m->f, residual, (size_t)T * D, 0, m->f, s))) return rc;
        if (c.parallel_attn || c.share_input)
            if ((rc = ifa_add(m->f, x, (size_t)T * D, 0, m->f, s))) return rc;
        if (f_post) {
            if ((rc = norm_rows(m, m->f, T, L.t[T_FFN_POST_NORM], L.t[T_FFN_POST_NORM_B], m->hn, 0.0f))) return rc;
            std::swap(m->f, m->hn);
        }
    }
    std::swap(m->x, m->f);
    x = m->x;
    return IFA_OK;
}

static bool batch_fused_ok(const ifa_model *m, int n);
static bool prefill_big_ok(const ifa_model *m);
// no_head: a chunk of a longer prompt that is not its last one -- the layers only (KV cache rows written), no lm_head / argmax / sync
static int forward_ops(ifa_model *m, const int *tokens_host, int T, int prefix_len, void *logits_out, int *next_token, bool no_head = false)
{
    const ifa_model_config &c = m->cfg;
    if (T <= 0 || prefix_len < 0 || prefix_len + T > c.max_ctx)
        return ifa_fail(IFA_ERR_ARG, "forward: %d tokens at prefix %d exceed max_ctx %d", T, prefix_len, c.max_ctx);
    int rc = ensure_scratch(m, T);
    if (rc) return rc;
    ifa_stream s = m->stream;
    const size_t D = c.dim, QD = (size_t)c.heads * c.head_dim, KVD = (size_t)c.kv_heads * c.head_dim;
    const ifa_tp_topology *tp = m->topo;            // multi-GPU partition (ifa_model_tp_prefill): merges + stage hand-over
    const bool first_stage = !tp || tp->stage == 0, last_stage = !tp || tp->next_rank < 0 || tp->n_stages == 1;
    const bool merging = tp_merging(m);
    if (first_stage && (!m->g[T_EMBD].present() || m->g[T_EMBD].dtype != F16)) return ifa_fail(IFA_ERR_STATE, "F16 embeddings not set");
    static const bool trace_host = getenv("IFA_TRACE_FORWARD") != nullptr;
    const auto host_t0 = std::chrono::steady_clock::now();
    if (first_stage) {
        IFA_HIP_CHECK(hipMemcpyAsync(m->tokens_dev, tokens_host, sizeof(int) * (size_t)T, hipMemcpyHostToDevice, m->stream));
        k_gather_rows<<<dim3(4, (unsigned)T), dim3(256), 0, m->stream>>>((const half_t *)m->g[T_EMBD].data, m->tokens_dev, T, (int)D,
                                                                          (int)m->g[T_EMBD].rows, m->x, c.embd_scale);
        IFA_LAUNCH_CHECK();
    } else if ((rc = ifa_recv(tp->world, m->x, (size_t)T * D * 2, tp->prev_rank, s))) return rc;     // the previous group's [T][dim] output
    half_t *x = m->x;
    const Tensor none;
    // small element-wise ops are one launch where the wiring allows it (each keeps its own half rounding): RoPE(q) + RoPE(k)
    // + the F16 cache rows; the residual Add + the norm that follows it (a layer is ~17 launches otherwise, and at short
    // prompts every one of them is a fixed ~5 us)
    const bool seq_wiring = !c.parallel_attn && !c.share_input;
    bool xn_ready = false;           // m->xn already holds the next norm's output (fused into the previous layer's last Add)
    // prompts of 2..16 tokens on a dense Q4 model with the sequential RMS wiring: the linears of a layer as FOUR launches of
    // the rows GEMM (ifa_gemm_rows_mfma.hip) -- norm prologue + wq | wk | wv into q / k / v, wo + residual, norm + w1 / w3 +
    // GLU, w2 + residual -- instead of seven products and four element-wise launches (9..16 tokens: the norms stay launches)
    // Prompts above `prefill_big_min` tokens (47; round 4: 128) take the same four launches per layer from the large-tile GEMM (ifa_gemm.hip, k_gemm_big: the
    // weights dequantised once per workgroup and step into LDS; reference-layout rows), norms as their own launches.
    const bool pf_big = !tp && T > std::max(32, m->opt_prefill_big_min) && prefill_big_ok(m);
    bool pf_fused = pf_big || (!tp && T >= 2 && T <= 32 && batch_fused_ok(m, T) && c.experts == 0);
    if (pf_fused && !pf_big) {
        if ((rc = ensure_mo(m))) return rc;
        pf_fused = batch_fused_ok(m, T);          // (ensure_mo may have switched the copies off: ask again, see forward_batch)
    }
    if (pf_big && (rc = ensure_x32(m))) return rc;
    // the mid-length kernel: every linear of every layer a 20-byte-block Q4 tensor with its tiled copy, dense FFN
    bool pf_mid = pf_big && m->opt_prefill_mid && m->opt_rows_mo && T <= m->opt_prefill_mid_max && c.experts == 0;
    if (pf_mid && (rc = ensure_mo(m))) return rc;
    for (int l = 0; l < c.layers && pf_mid; l++) {
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) { const Tensor &t = m->layers[(size_t)l].t[id]; if (!t.present() || !t.mo || t.cols % 128 != 0 || t.rows % 16 != 0) pf_mid = false; }
    }
    for (int l = 0; l < c.layers && pf_fused; l++) {
        Layer &L = m->layers[l];
        const size_t F = c.ffn;
        const bool norm_fused = !pf_big && (T <= 8 || rows_mo(m, L.t[T_WQ])) && T <= 16;
        auto wp = [&](int id) { return pf_mid ? (const uint8_t *)L.t[id].mo : (pf_big ? (const uint8_t *)(L.t[id].x32 ? L.t[id].x32 : L.t[id].data) : rows_w(m, L.t[id])); };
        const int mo_flag = pf_mid ? 1 : (pf_big ? 0 : rows_mo(m, L.t[T_WQ]));
        auto lin = [&](const GmArgs &A, int id, int epi, int norm) {
            if (pf_mid && gemm_mid_ok(L.t[id].dtype, A, epi)) return gemm_mid(A, epi, m->stream);
            if (pf_mid) return ifa_fail(IFA_ERR_STATE, "mid-length GEMM declined a product of layer tensor %d", id);
            return pf_big ? gemm_big(L.t[id].x32 ? (int)Q4_B32T1A : L.t[id].dtype, A, epi, m->stream) : gemm_rows_mfma_launch(A, epi, norm, m->stream);
        };
        Tensor nob;
        GmArgs P;
        auto clear = [&]() { memset(&P, 0, sizeof(P)); P.T = T; P.eps = c.eps; P.act_kind = c.act_kind; P.mo = mo_flag; P.no_waits = pf_big ? !m->opt_gemm_splitk : !m->opt_rows_kparts; };
        clear();
        if (!norm_fused && (rc = norm_rows(m, x, T, L.t[T_ATTN_NORM], pf_big ? L.t[T_ATTN_NORM_B] : nob, m->xn, c.attn_norm_base))) return rc;
        P.W[0] = wp(T_WQ); P.W[1] = wp(T_WK); P.W[2] = wp(T_WV);
        P.rows[0] = (int)QD; P.rows[1] = (int)KVD; P.rows[2] = (int)KVD; P.nsets = 3; P.nblk = (int)(D / 32);
        P.X = norm_fused ? x : m->xn; P.ldx = (int)D;
        if (norm_fused) { P.norm_w = (const half_t *)L.t[T_ATTN_NORM].data; P.multi_base = c.attn_norm_base; }
        P.bias[0] = (const half_t *)L.t[T_WQ_B].data; P.bias[1] = (const half_t *)L.t[T_WK_B].data; P.bias[2] = (const half_t *)L.t[T_WV_B].data;
        P.Yset[0] = m->q; P.Yset[1] = m->k; P.Yset[2] = m->v; P.ldyset[0] = (int)QD; P.ldyset[1] = (int)KVD; P.ldyset[2] = (int)KVD;
        if ((rc = lin(P, T_WQ, GM_PLAIN, norm_fused ? 1 : 0))) return rc;
        uint8_t *kdst = (uint8_t *)L.kcache + (size_t)prefix_len * m->kv_row_bytes;
        uint8_t *vdst = (uint8_t *)L.vcache + (size_t)prefix_len * m->kv_row_bytes;
        const bool kv_f16 = c.kv_dtype != Q8_B32T2;
        bool kv_stored = false;
        if (c.rope_order != 0) {
            rc = ifa_rope_qk_store(m->q, m->k, m->v, c.head_dim, c.heads, c.kv_heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary,
                                   kv_f16 ? kdst : nullptr, kv_f16 ? vdst : nullptr, m->kv_row_bytes / 2, s);
            if (rc == IFA_OK) kv_stored = kv_f16;
            else if (rc != IFA_ERR_STATE) return rc;
            else {
                if ((rc = ifa_rope(m->q, c.head_dim, c.heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
                if ((rc = ifa_rope(m->k, c.head_dim, c.kv_heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
            }
        }
        if (!kv_f16) {
            if ((rc = ifa_quantize_act_q8(m->k, T, KVD, kdst, s))) return rc;
            if ((rc = ifa_quantize_act_q8(m->v, T, KVD, vdst, s))) return rc;
        } else if (!kv_stored) {
            IFA_HIP_CHECK(hipMemcpyAsync(kdst, m->k, (size_t)T * m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
            IFA_HIP_CHECK(hipMemcpyAsync(vdst, m->v, (size_t)T * m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
        }
        if ((rc = ifa_attention(m->q, L.kcache, L.vcache, c.kv_dtype, prefix_len + T, T, prefix_len, c.heads, c.kv_heads,
                                c.head_dim, c.use_alibi ? 1.0f : c.kq_scale, c.use_alibi, c.tp_rank * c.heads,
                                c.heads * std::max(1, c.tp_size), m->att, s))) return rc;
        clear();
        P.W[0] = wp(T_WO); P.rows[0] = (int)D; P.nsets = 1; P.nblk = (int)(QD / 32);
        P.X = m->att; P.ldx = (int)QD; P.bias[0] = (const half_t *)L.t[T_WO_B].data;
        P.Y = m->a; P.ldy = (int)D; P.res = x; P.ldres = (int)D;
        if ((rc = lin(P, T_WO, GM_RESIDUAL, 0))) return rc;
        clear();
        if (!norm_fused && (rc = norm_rows(m, m->a, T, L.t[T_FFN_NORM], pf_big ? L.t[T_FFN_NORM_B] : nob, m->hn, c.ffn_norm_base))) return rc;
        if (c.experts > 0 && L.t[T_MOE_GATE].present()) {      // (pf_big only) mixture of experts: the attention half fused, the expert FFNs device-routed
            if ((rc = moe_ffn(m, L, m->hn, T))) return rc;
            if ((rc = ifa_add(m->f, m->a, (size_t)T * D, 0, m->f, s))) return rc;
            std::swap(m->x, m->f);
            x = m->x;
            continue;
        }
        P.W[0] = wp(T_W1); P.W1 = wp(T_W3); P.rows[0] = (int)F; P.nsets = 1; P.nblk = (int)(D / 32);
        P.X = norm_fused ? m->a : m->hn; P.ldx = (int)D;
        if (norm_fused) { P.norm_w = (const half_t *)L.t[T_FFN_NORM].data; P.multi_base = c.ffn_norm_base; }
        P.bias[0] = (const half_t *)L.t[T_W1_B].data; P.bias1 = (const half_t *)L.t[T_W3_B].data;
        P.Y = m->t1; P.ldy = (int)F;
        if ((rc = lin(P, T_W1, GM_GLU, norm_fused ? 1 : 0))) return rc;
        clear();
        P.W[0] = wp(T_W2); P.rows[0] = (int)D; P.nsets = 1; P.nblk = (int)(F / 32);
        P.X = m->t1; P.ldx = (int)F; P.bias[0] = (const half_t *)L.t[T_W2_B].data;
        P.Y = m->f; P.ldy = (int)D; P.res = m->a; P.ldres = (int)D;
        if ((rc = lin(P, T_W2, GM_RESIDUAL, 0))) return rc;
        std::swap(m->x, m->f);
        x = m->x;
    }
    for (int l = pf_fused ? c.layers : 0; l < c.layers; l++) {
        Layer &L = m->layers[l];
        const half_t *attn_in = x;
        if (L.t[T_ATTN_NORM].present()) {
            if (!xn_ready && (rc = norm_rows(m, x, T, L.t[T_ATTN_NORM], L.t[T_ATTN_NORM_B], m->xn, c.attn_norm_base))) return rc;
            attn_in = m->xn;
        }
        xn_ready = false;
        if ((rc = matmul(m, attn_in, T, L.t[T_WQ], L.t[T_WQ_B], m->q))) return rc;
        if ((rc = matmul(m, attn_in, T, L.t[T_WK], L.t[T_WK_B], m->k))) return rc;
        if ((rc = matmul(m, attn_in, T, L.t[T_WV], L.t[T_WV_B], m->v))) return rc;
        uint8_t *kdst = (uint8_t *)L.kcache + (size_t)prefix_len * m->kv_row_bytes;
        uint8_t *vdst = (uint8_t *)L.vcache + (size_t)prefix_len * m->kv_row_bytes;
        const bool kv_f16 = c.kv_dtype != Q8_B32T2;
        bool kv_stored = false;
        if (c.rope_order != 0) {
            rc = ifa_rope_qk_store(m->q, m->k, m->v, c.head_dim, c.heads, c.kv_heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary,
                                   kv_f16 ? kdst : nullptr, kv_f16 ? vdst : nullptr, m->kv_row_bytes / 2, s);
            if (rc == IFA_OK) kv_stored = kv_f16;
            else if (rc != IFA_ERR_STATE) return rc;
            else {
                if ((rc = ifa_rope(m->q, c.head_dim, c.heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
                if ((rc = ifa_rope(m->k, c.head_dim, c.kv_heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
            }
        }
        if (!kv_f16) {
            if ((rc = ifa_quantize_act_q8(m->k, T, KVD, kdst, s))) return rc;
            if ((rc = ifa_quantize_act_q8(m->v, T, KVD, vdst, s))) return rc;
        } else if (!kv_stored) {
            IFA_HIP_CHECK(hipMemcpyAsync(kdst, m->k, (size_t)T * m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
            IFA_HIP_CHECK(hipMemcpyAsync(vdst, m->v, (size_t)T * m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
        }
        if ((rc = ifa_attention(m->q, L.kcache, L.vcache, c.kv_dtype, prefix_len + T, T, prefix_len, c.heads, c.kv_heads,
                                c.head_dim, c.use_alibi ? 1.0f : c.kq_scale, c.use_alibi, c.tp_rank * c.heads,
                                c.heads * std::max(1, c.tp_size), m->att, s))) return rc;
        if ((rc = matmul(m, m->att, T, L.t[T_WO], merging ? none : L.t[T_WO_B], m->a))) return rc;
        if ((rc = tp_merge_rows(m, m->a, T, L.t[T_WO_B]))) return rc;       // BY_TENSOR: sum of the ranks' partial products, bias after
        if ((rc = layer_tail_ops(m, l, T, x, attn_in, xn_ready))) return rc;
    }
    if (!last_stage) {       // BY_LAYER / HYBRID: hand the [T][dim] output to the next device group, then learn the token
        if ((rc = ifa_send(tp->world, x, (size_t)T * D * 2, tp->next_rank, s))) return rc;
        if ((rc = tp_argmax_scratch(m, 1))) return rc;
        if ((rc = ifa_broadcast(tp->world, m->tp_tok, 4, tp->token_src, s))) return rc;
        IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned, m->tp_tok, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
        if (next_token) *next_token = m->host_pinned[0];
        return IFA_OK;
    }
    if (no_head && !logits_out) return IFA_OK;
    if (scale_on(c.out_scale) && (rc = ifa_scale(x, c.out_scale, (size_t)T * D, x, s))) return rc;
    const half_t *hfin = x;
    if (m->g[T_OUT_NORM].present()) {
        if (!xn_ready && (rc = norm_rows(m, x, T, m->g[T_OUT_NORM], m->g[T_OUT_NORM_B], m->xn, c.out_norm_base))) return rc;
        hfin = m->xn;
    } else {
        IFA_HIP_CHECK(hipMemcpyAsync(m->xn, x, (size_t)T * D * 2, hipMemcpyDeviceToDevice, m->stream));
    }
    const Tensor &lm = m->g[T_LM_HEAD];
    const size_t V = lm.rows;                        // (this rank's vocabulary shard under tensor parallelism)
    int t0 = logits_out ? 0 : T - 1;
    if (logits_out) { if ((rc = matmul(m, hfin, T, lm, none, m->logits))) return rc; }
    else if (lm.dtype == F16 && D % 8 == 0 && D <= 8192) {
        // the last row only: the decode step's lm_head kernel on the normalised row (same per-row chain as the op-level GEMV --
        // bit-identical logits -- at 6 TB/s instead of 1.1: 232 -> 45 us per prompt, rocprofv3 r06)
        DecLmHeadParams H2; memset(&H2, 0, sizeof(H2));
        H2.x = hfin + (size_t)t0 * D; H2.eps = c.eps; H2.cols = (int)D; H2.W = (const half_t *)lm.data; H2.logits = m->logits + (size_t)t0 * V; H2.rows = (int)V;
        if ((rc = launch_lmhead(H2, 0, m->opt_rpw_lm, m->stream))) return rc;
    }
    else { if ((rc = matmul(m, hfin + (size_t)t0 * D, 1, lm, none, m->logits + (size_t)t0 * V))) return rc; }
    if (logits_out) IFA_HIP_CHECK(hipMemcpyAsync(logits_out, m->logits, (size_t)T * V * 2, hipMemcpyDeviceToDevice, m->stream));
    if (tp) {                // distributed argmax of the last row over the group's shards (+ announcement to the other groups)
        if ((rc = tp_pick_rows(m, *tp, m->logits + (size_t)(T - 1) * V, V, (int)V, 1))) return rc;
        if (tp->n_stages > 1 && (rc = ifa_broadcast(tp->world, m->tp_tok, 4, tp->token_src, s))) return rc;
        IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned, m->tp_tok, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    } else {
        if ((rc = ifa_argmax_masked(m->logits + (size_t)(T - 1) * V, V, m->state + 3, m->state, s))) return rc;
        IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned, m->state, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    }
    const auto host_t1 = std::chrono::steady_clock::now();
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    if ((rc = wait_err_check("forward step"))) { drop_graphs(m); return rc; }      // (a split-K / K-parts wait gave up: the step is not valid; those launches are off now)
    if (trace_host)      // how much of a step is the host enqueuing (launch-bound) vs the GPU draining what was enqueued
        fprintf(stderr, "forward T=%d: enqueue %.3f ms, total %.3f ms\n", T, std::chrono::duration<double, std::milli>(host_t1 - host_t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count());
    if (next_token) *next_token = m->host_pinned[0];
    return IFA_OK;
}

// ------------------------------------------------ dynamic batching: one new token for each of n queries
// (QueryStateTable + Infer_Std over several queries, src/transformer/inference_engine.cc:1054-1220): the linear layers
// run once over the n rows (weights streamed once: MFMA GEMM), RoPE / KV store / attention per row on the KV cache
// set of its query.
struct AttnRowH { const void *kc, *vc; int n_ctx, pad; };
extern "C" int ifa_attention_rows(const void *q, const void *rows_dev, int kv_dtype, int n_rows, int max_ctx, int heads, int kv_heads,
                                  int head_dim, float kq_scale, int alibi, int alibi_base_head, int alibi_total_heads, void *out,
                                  ifa_stream stream);
extern "C" int ifa_rope_rows(void *x, int head_dim, int heads, int tokens, const int *positions_dev, float theta, int order,
                             float partial_rotary_factor, ifa_stream stream);

// row r of k / v -> position rows[r].n_ctx - 1 of its query's cache (F16 copy or Q8_B32T2 quantisation, 32 lanes per block)
template <bool Q8>
__global__ void __launch_bounds__(256) k_kv_store_rows(const half_t *__restrict__ k, const half_t *__restrict__ v, int kv_dim,
                                                       size_t row_bytes, const AttnRowH *__restrict__ rows)
{
    const int r = blockIdx.x, which = blockIdx.y;
    const half_t *src = (which ? v : k) + (size_t)r * kv_dim;
    uint8_t *dst = (uint8_t *)(which ? rows[r].vc : rows[r].kc) + (size_t)(rows[r].n_ctx - 1) * row_bytes;
    if constexpr (!Q8) {
        for (int c = threadIdx.x; c < kv_dim; c += 256) reinterpret_cast<half_t *>(dst)[c] = src[c];
    } else {
        const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
        for (int b = grp; b < kv_dim / 32; b += 8) {      // Tensor_QuantizeQ8_B32T2_Alg2_Kernel (tensor_quant.h:44-82)
            const float val = h2f(src[b * 32 + lane]);
            float mx = fabsf(val);
#pragma unroll
            for (int m2 = 16; m2 > 0; m2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m2, 32));
            const float sc = mx / 127;
            int qv = sc <= 0.000001f ? 0 : (int)roundf(val / sc);
            qv = min(max(qv, -128), 127);
            uint8_t *blk = dst + (size_t)b * 34;
            blk[2 + lane] = (uint8_t)(int8_t)qv;
            if (lane == 0) *reinterpret_cast<uint16_t *>(blk) = __builtin_bit_cast(uint16_t, f2h(sc));
        }
    }
}

static void *kv_ptr(ifa_model *m, size_t layer, int slot, bool is_v)
{
    if (slot == m->cur_slot || m->slots.empty()) return is_v ? m->layers[layer].vcache : m->layers[layer].kcache;
    const ifa_model::KvSlot &sl = m->slots[(size_t)slot];
    return is_v ? sl.v[layer] : sl.k[layer];
}


// The rows GEMM's own copy of the seven matrices of every dense layer (MO layout): built when the first batched step or short
// prompt needs it (never inside a stream capture), dropped with the tensor.  4.3 GB more for Llama-2-7B Q4.
static int ensure_mo_build(ifa_model *m);
// (a failed allocation -- the copies are 4.3 GB for Llama-2-7B Q4 -- is not an error of the step that triggered it: every partial
//  copy is dropped, opt_rows_mo goes off and the tiled kernels (<= 16 rows per launch) serve the batched steps)
static int ensure_mo(ifa_model *m)
{
    if (!m->opt_rows_mo) return IFA_OK;
    // (option "debug_mo_alloc_fail": the build reports an allocation failure after its first copy -- the tests' way to walk the downgrade)
    const int rc = ensure_mo_build(m);
    if (rc == IFA_OK) return IFA_OK;
    (void)hipGetLastError();
    for (Layer &L : m->layers) {
        for (Tensor &t : L.t) if (t.mo) { (void)hipFree(t.mo); t.mo = nullptr; }
        for (Tensor &t : L.experts) if (t.mo) { (void)hipFree(t.mo); t.mo = nullptr; }
        if (L.moe_table_mo) { (void)hipFree(L.moe_table_mo); L.moe_table_mo = nullptr; }
    }
    m->opt_rows_mo = 0;
    drop_graphs(m);
    fprintf(stderr, "inferflow_amd: the rows GEMM's operand-order weight copies could not be built (%s); batched steps use the tiled kernels\n", ifa_last_error());
    return IFA_OK;
}
static int ensure_mo_build(ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    bool built = false;
    for (Layer &L : m->layers) {
        const bool moe = c.experts > 0 && L.t[T_MOE_GATE].present();
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) {
            Tensor &t = L.t[id];
            if (moe && (id == T_W1 || id == T_W3 || id == T_W2)) continue;
            if (t.mo || !t.present() || !t.tiled || !rows_mo_fmt(t.dtype) || t.cols % 128 != 0) continue;
            if (m->opt_debug_mo_alloc_fail && built) return ifa_fail(IFA_ERR_HIP, "hipMalloc: out of memory (simulated: debug_mo_alloc_fail)");
            IFA_HIP_CHECK(hipMalloc(&t.mo, gemm_rows_mo_bytes(t.rows, t.cols)));
            int rc = gemm_rows_mo_build(t.dtype, t.tiled, t.rows, t.cols, t.mo, m->stream);
            if (rc) return rc;
            built = true;
        }
    }
    // mixture-of-experts layers: the experts' matrices too, with their pointer table {w1, w3, w2, -} (the order of moe_table):
    // the experts that collect 2..8 rows of a batched step take the grouped MO launches (moe_ffn_device)
    for (Layer &L : m->layers) {
        if (!(c.experts > 0 && L.t[T_MOE_GATE].present()) || L.moe_table_mo || (int)L.experts.size() != c.experts * 3) continue;
        bool ok = true;
        for (Tensor &t : L.experts) ok = ok && t.present() && t.tiled && is_q4(t.dtype) && t.cols % 128 == 0;
        if (!ok) continue;
        std::vector<void *> tab((size_t)c.experts * 4, nullptr);
        for (int e = 0; e < c.experts; e++) {
            const int order[3] = {0, 2, 1};                       // experts[e * 3 + {0, 1, 2}] = w1, w2, w3
            for (int k = 0; k < 3; k++) {
                Tensor &t = L.experts[(size_t)e * 3 + order[k]];
                if (!t.mo) {
                    IFA_HIP_CHECK(hipMalloc(&t.mo, gemm_rows_mo_bytes(t.rows, t.cols)));
                    int rc = gemm_rows_mo_build(t.dtype, t.tiled, t.rows, t.cols, t.mo, m->stream);
                    if (rc) return rc;
                    built = true;
                }
                tab[(size_t)e * 4 + k] = t.mo;
            }
        }
        IFA_HIP_CHECK(hipMalloc(&L.moe_table_mo, tab.size() * sizeof(void *)));
        IFA_HIP_CHECK(hipMemcpy(L.moe_table_mo, tab.data(), tab.size() * sizeof(void *), hipMemcpyHostToDevice));
    }
    if (built) IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    return IFA_OK;
}
// Long prompts of a model in a 64-weight nibble format: Q4_B32T1A-layout copies of the dense layers' matrices for k_gemm_big
static int ensure_x32(ifa_model *m)
{
    bool built = false;
    for (Layer &L : m->layers) {
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) {
            Tensor &t = L.t[id];
            if (t.x32 || !t.present() || !t.tiled || (t.dtype != Q4_B64T1 && t.dtype != Q3H_B64T1) || t.cols % 64 != 0) continue;
            IFA_HIP_CHECK(hipMalloc(&t.x32, t.rows * (t.cols / 32) * 20));
            int rc = expand_b64_to_q4b32(t.dtype, t.tiled, t.rows, t.cols, t.x32, m->stream);
            if (rc) return rc;
            built = true;
        }
    }
    if (built) IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    return IFA_OK;
}
// weight pointer of a rows-GEMM launch: the MO copy when it exists (all sets of a launch alike: ensure_mo builds all or none)
static const uint8_t *rows_w(const ifa_model *m, const Tensor &t) { return (const uint8_t *)(m->opt_rows_mo && t.mo ? t.mo : t.tiled); }
static int rows_mo(const ifa_model *m, const Tensor &t) { return m->opt_rows_mo && t.mo ? 1 : 0; }

// ---- the batched step as five launches per layer (the structure of the batch-1 step: ifa_gemm_rows_mfma.hip with the norm
// prologue / GLU / residual epilogues, k_dec_attn<.., BATCH>): dense models with the sequential RMS wiring, every linear in
// tiled Q4_B32T1, 2..16 queries.  Everything else takes the op-by-op rows below.
static bool batch_fused_ok(const ifa_model *m, int n)
{
    const ifa_model_config &c = m->cfg;
    if (!m->opt_batch_fused || !m->opt_gemm_rows || !gemm_rows_use_mfma() || n < 2 || n > (m->opt_rows_mo ? 32 : 16) || m->topo) return false;      // (17..32 rows: MO copies only)
    if (has_post_norms(m)) return false;
    if (c.norm_kind != 0 || c.parallel_attn || c.share_input) return false;
    if (scale_on(c.attn_out_scale) || scale_on(c.ffn_out_scale) || scale_on(c.out_scale)) return false;
    const size_t D = c.dim, QD = (size_t)c.heads * c.head_dim, KVD = (size_t)c.kv_heads * c.head_dim, F = c.ffn;
    if (D % 128 || QD % 128 || F % 128 || D > 4096 || KVD % 16 || D % 16 || F % 16) return false;
    if (c.head_dim != 32 && c.head_dim != 48 && c.head_dim != 64 && c.head_dim != 80 && c.head_dim != 96 && c.head_dim != 128) return false;
    if (c.kv_dtype == Q8_B32T2 && c.head_dim % 32 != 0) return false;
    if (dec_attn_smem(c.head_dim, c.max_ctx) > IFA_LDS_LIMIT) return false;
    for (const Layer &L : m->layers) {
        const bool moe = c.experts > 0 && L.t[T_MOE_GATE].present();      // MoE layers: the attention half is fused, the FFN runs moe_ffn
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) {
            if (moe && (id == T_W1 || id == T_W3 || id == T_W2)) continue;
            if (!L.t[id].present() || !L.t[id].tiled) return false;
            if (!is_q4(L.t[id].dtype) && !(m->opt_rows_mo && rows_mo_fmt(L.t[id].dtype) && L.t[id].cols % 128 == 0)) return false;
        }
        if (moe && !moe_device_ok(m, L)) return false;
        if (!L.t[T_ATTN_NORM].present() || !L.t[T_FFN_NORM].present() || L.t[T_ATTN_NORM_B].present() || L.t[T_FFN_NORM_B].present()) return false;
    }
    return true;
}

// prompts above `prefill_big_min` tokens as four launches of the large-tile GEMM per layer (forward_ops, pf_big): dense layers with the
// sequential wiring, every linear a 20-byte-block Q4 tensor (wq / wk / wv of one format), dims in multiples of 64
static bool prefill_big_ok(const ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    if (!m->opt_prefill_big || m->topo || c.parallel_attn || c.share_input || has_post_norms(m)) return false;
    if (scale_on(c.attn_out_scale) || scale_on(c.ffn_out_scale)) return false;
    const size_t D = c.dim, QD = (size_t)c.heads * c.head_dim, F = c.ffn;
    if (D % 64 || QD % 64 || F % 64) return false;
    for (const Layer &L : m->layers) {
        const bool moe = c.experts > 0 && L.t[T_MOE_GATE].present();      // MoE layers: the attention half is fused, the FFN runs moe_ffn
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) {
            if (moe && (id == T_W1 || id == T_W3 || id == T_W2)) continue;
            const bool b64 = (L.t[id].dtype == Q4_B64T1 || L.t[id].dtype == Q3H_B64T1) && L.t[id].tiled;      // via their Q4_B32T1A-layout copy (ensure_x32)
            if (!L.t[id].present() || !L.t[id].data || (L.t[id].dtype != Q4_B32T1A && L.t[id].dtype != Q4_B32T1B && !b64)) return false;
        }
        if (L.t[T_WK].dtype != L.t[T_WQ].dtype || L.t[T_WV].dtype != L.t[T_WQ].dtype || (!moe && L.t[T_W3].dtype != L.t[T_W1].dtype)) return false;
        if (!L.t[T_ATTN_NORM].present() || !L.t[T_FFN_NORM].present()) return false;
    }
    return true;
}

static int batch_fused_layer(ifa_model *m, int l, int n, const half_t *x, half_t *xnext, const void *rows_l)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    const size_t D = c.dim, QD = (size_t)c.heads * c.head_dim, KVD = (size_t)c.kv_heads * c.head_dim, F = c.ffn;
    int rc;
    GmArgs P;
    auto clear = [&]() { memset(&P, 0, sizeof(P)); P.T = n; P.eps = c.eps; P.act_kind = c.act_kind; P.no_waits = !m->opt_rows_kparts; };
    // 1. RmsNorm -> wq | wk | wv  (one virtual row space, one [n][q | k | v] output)
    clear();
    P.W[0] = rows_w(m, L.t[T_WQ]); P.W[1] = rows_w(m, L.t[T_WK]); P.W[2] = rows_w(m, L.t[T_WV]); P.mo = rows_mo(m, L.t[T_WQ]);
    P.rows[0] = (int)QD; P.rows[1] = (int)KVD; P.rows[2] = (int)KVD; P.nsets = 3; P.nblk = (int)(D / 32);
    // (9..16 queries: the activation rows are staged in chunks of 2048 columns, so the norm runs as its own launch)
    const bool norm_fused = (n <= 8 || rows_mo(m, L.t[T_WQ])) && n <= 16;      // (MO layout: 16 rows x 4096 columns are one chunk too; 17..32 rows: chunked)
    Tensor nob;
    if (!norm_fused && (rc = norm_rows(m, x, n, L.t[T_ATTN_NORM], nob, m->xn, c.attn_norm_base))) return rc;
    P.X = norm_fused ? x : m->xn; P.ldx = (int)D;
    if (norm_fused) { P.norm_w = (const half_t *)L.t[T_ATTN_NORM].data; P.multi_base = c.attn_norm_base; }
    P.bias[0] = (const half_t *)L.t[T_WQ_B].data; P.bias[1] = (const half_t *)L.t[T_WK_B].data; P.bias[2] = (const half_t *)L.t[T_WV_B].data;
    P.Y = m->bqkv; P.ldy = (int)(QD + 2 * KVD);
    if ((rc = gemm_rows_mfma_launch(P, GM_PLAIN, norm_fused ? 1 : 0, m->stream))) return rc;
    // 2. RoPE, KV store, attention of every query on its own cache
    {
        const int rope_dims = (int)(c.head_dim * c.partial_rotary + 0.5f);
        DecAttnParams A; memset(&A, 0, sizeof(A));
        A.q = m->bqkv; A.k_new = m->bqkv + QD; A.v_new = A.k_new + KVD;
        A.state = m->state; A.rope_tab = m->brope; A.heads = c.heads; A.kv_heads = c.kv_heads;
        A.kv_q8 = c.kv_dtype == Q8_B32T2; A.kq_scale = c.use_alibi ? 1.0f : c.kq_scale;
        A.rope_order = c.rope_order; A.rope_cols = rope_dims;
        A.alibi = c.use_alibi; A.alibi_base = c.tp_rank * c.heads; A.alibi_total = c.heads * std::max(1, c.tp_size);
        A.out = m->att; A.max_ctx = c.max_ctx; A.batch_rows = rows_l; A.q_stride = (int)(QD + 2 * KVD);
        const size_t asmem = dec_attn_smem(c.head_dim, c.max_ctx);
        const dim3 grid((unsigned)c.heads, (unsigned)n), block(256);
#define IFA_BATTN(HDV, Q8V) { auto kern = k_dec_attn<HDV, Q8V, true>; \
        if (asmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)asmem)); \
        kern<<<grid, block, asmem, m->stream>>>(A.q, nullptr, nullptr, A.heads, A.kv_heads, A); }
        switch (c.head_dim) {
        case 32: if (A.kv_q8) IFA_BATTN(32, true) else IFA_BATTN(32, false) break;
        case 64: if (A.kv_q8) IFA_BATTN(64, true) else IFA_BATTN(64, false) break;
        case 96: if (A.kv_q8) IFA_BATTN(96, true) else IFA_BATTN(96, false) break;
        case 128: if (A.kv_q8) IFA_BATTN(128, true) else IFA_BATTN(128, false) break;
        case 48: IFA_BATTN(48, false) break;
        case 80: IFA_BATTN(80, false) break;
        default: return ifa_fail(IFA_ERR_ARG, "fused batched attention: head_dim %d", c.head_dim);
        }
#undef IFA_BATTN
        IFA_LAUNCH_CHECK();
    }
    // 3. wo (+ bias) + residual
    clear();
    P.W[0] = rows_w(m, L.t[T_WO]); P.mo = rows_mo(m, L.t[T_WO]); P.rows[0] = (int)D; P.nsets = 1; P.nblk = (int)(QD / 32);
    P.X = m->att; P.ldx = (int)QD; P.bias[0] = (const half_t *)L.t[T_WO_B].data;
    P.Y = m->a; P.ldy = (int)D; P.res = x; P.ldres = (int)D;
    if ((rc = gemm_rows_mfma_launch(P, GM_RESIDUAL, 0, m->stream))) return rc;
    if (c.experts > 0 && L.t[T_MOE_GATE].present()) {
        // mixture of experts: norm, the device-routed expert FFNs over the n rows (moe_ffn_device), residual
        Tensor none;
        if (moe_device_ok(m, L) && moe_router_rows_ok(m, L, n)) return moe_ffn_device(m, L, m->hn, n, m->a, m->a, xnext);
        {
            if ((rc = norm_rows(m, m->a, n, L.t[T_FFN_NORM], none, m->hn, c.ffn_norm_base))) return rc;
            if ((rc = moe_ffn(m, L, m->hn, n))) return rc;
        }
        return ifa_add(m->f, m->a, (size_t)n * D, 0, xnext, (ifa_stream)m->stream);
    }
    // 4. RmsNorm -> w1, w3 -> act(w1 x) * (w3 x)
    clear();
    P.W[0] = rows_w(m, L.t[T_W1]); P.W1 = rows_w(m, L.t[T_W3]); P.mo = rows_mo(m, L.t[T_W1]); P.rows[0] = (int)F; P.nsets = 1; P.nblk = (int)(D / 32);
    if (!norm_fused && (rc = norm_rows(m, m->a, n, L.t[T_FFN_NORM], nob, m->hn, c.ffn_norm_base))) return rc;
    P.X = norm_fused ? m->a : m->hn; P.ldx = (int)D;
    if (norm_fused) { P.norm_w = (const half_t *)L.t[T_FFN_NORM].data; P.multi_base = c.ffn_norm_base; }
    P.bias[0] = (const half_t *)L.t[T_W1_B].data; P.bias1 = (const half_t *)L.t[T_W3_B].data;
    P.Y = m->t1; P.ldy = (int)F;
    if ((rc = gemm_rows_mfma_launch(P, GM_GLU, norm_fused ? 1 : 0, m->stream))) return rc;
    // 5. w2 (+ bias) + residual -> the next layer's input
    clear();
    P.W[0] = rows_w(m, L.t[T_W2]); P.mo = rows_mo(m, L.t[T_W2]); P.rows[0] = (int)D; P.nsets = 1; P.nblk = (int)(F / 32);
    P.X = m->t1; P.ldx = (int)F; P.bias[0] = (const half_t *)L.t[T_W2_B].data;
    P.Y = xnext; P.ldy = (int)D; P.res = m->a; P.ldres = (int)D;
    return gemm_rows_mfma_launch(P, GM_RESIDUAL, 0, m->stream);
}

static int forward_batch(ifa_model *m, int n, const int *tokens_host, const int *pos_host, const int *slot_host, int *next_tokens,
                         void *logits_out)
{
    const ifa_model_config &c = m->cfg;
    const int n_slots = m->slots.empty() ? 1 : (int)m->slots.size();
    int max_ctx = 0;
    for (int r = 0; r < n; r++) {
        if (pos_host[r] < 0 || pos_host[r] >= c.max_ctx) return ifa_fail(IFA_ERR_ARG, "decode_batch: position %d outside max_ctx %d", pos_host[r], c.max_ctx);
        if (slot_host[r] < 0 || slot_host[r] >= n_slots) return ifa_fail(IFA_ERR_ARG, "decode_batch: KV slot %d of %d", slot_host[r], n_slots);
        for (int r2 = 0; r2 < r; r2++) if (slot_host[r2] == slot_host[r]) return ifa_fail(IFA_ERR_ARG, "decode_batch: KV slot %d used twice", slot_host[r]);
        max_ctx = std::max(max_ctx, pos_host[r] + 1);
    }
    int rc = ensure_scratch(m, n);
    if (rc) return rc;
    ifa_stream s = m->stream;
    const int T = n;
    const size_t D = c.dim, KVD = (size_t)c.kv_heads * c.head_dim, L_ = m->layers.size();
    if (!m->g[T_EMBD].present() || m->g[T_EMBD].dtype != F16) return ifa_fail(IFA_ERR_STATE, "F16 embeddings not set");
    const ifa_tp_topology *tp = m->topo;            // tensor-parallel group (ifa_model_tp_decode_batch): merges + distributed argmax
    if (tp && tp->n_stages > 1) return ifa_fail(IFA_ERR_ARG, "decode_batch: layer groups are not batched (tensor-parallel groups only)");
    const bool merging = tp_merging(m);
    // per-step tables: positions, and for every layer the (k cache, v cache, context) of each row's query
    const size_t tab_bytes = L_ * (size_t)n * sizeof(AttnRowH) + 2 * (size_t)n * sizeof(int);
    if (tab_bytes > m->batch_tab_bytes) {
        drop_graphs(m);                            // the captured steps hold the old table addresses
        if (m->batch_tab_dev) IFA_HIP_CHECK(hipFree(m->batch_tab_dev));
        if (m->batch_tab_pin) IFA_HIP_CHECK(hipHostFree(m->batch_tab_pin));
        IFA_HIP_CHECK(hipMalloc(&m->batch_tab_dev, tab_bytes));
        IFA_HIP_CHECK(hipHostMalloc(&m->batch_tab_pin, tab_bytes, hipHostMallocDefault));
        m->batch_tab_bytes = tab_bytes;
    }
    AttnRowH *rows_h = (AttnRowH *)m->batch_tab_pin;
    int *pos_pin = (int *)(rows_h + L_ * (size_t)n);
    for (size_t l = 0; l < L_; l++)
        for (int r = 0; r < n; r++) {
            AttnRowH &a = rows_h[l * (size_t)n + r];
            a.kc = kv_ptr(m, l, slot_host[r], false); a.vc = kv_ptr(m, l, slot_host[r], true); a.n_ctx = pos_host[r] + 1; a.pad = 0;
        }
    for (int r = 0; r < n; r++) { pos_pin[r] = pos_host[r]; pos_pin[n + r] = tokens_host[r]; }
    const AttnRowH *rows_d = (const AttnRowH *)m->batch_tab_dev;
    const int *pos_d = (const int *)(rows_d + L_ * (size_t)n);
    const int *tok_d = pos_d + n;
    // Everything the device does in a step depends on the step only through the tables above (fixed addresses), so
    // for dense models the whole step -- table upload included -- is captured once per batch size and replayed.
    bool has_moe = false;
    for (const Layer &Lc : m->layers) has_moe = has_moe || (c.experts > 0 && Lc.t[T_MOE_GATE].present());
    // (measured on Llama-2-7B Q4: the batched step is bound by the small-T GEMM kernels, ~6.7 ms with or without the
    //  graph, so replay is opt-in: set_option("batch_graph", 1))
    bool fused = batch_fused_ok(m, n);          // five launches per layer: launch-bound without a graph, so it is replayed
    if (fused) {
        int rcm = ensure_mo(m); if (rcm) return rcm;
        // ensure_mo may have DOWNGRADED the model (the copies did not fit: opt_rows_mo = 0): what batch_fused_ok answered with the
        // copies in view -- up to 32 rows, the 64-weight formats -- no longer holds, so it is asked again before a path or a graph
        // is chosen; a step the tiled kernels do not cover takes the op-by-op rows below
        fused = batch_fused_ok(m, n);
        if (fused && (rcm = gemm_rows_kparts_reserve(m->stream))) return rcm;
    }
    if (has_moe && m->opt_moe_overlap) { int rcs = ensure_side_stream(m); if (rcs) return rcs; }
    // (MoE layers of the fused step route on the device -- no host round trip -- so they are captured too)
    const bool use_graph = (m->opt_batch_graph || fused) && m->opt_graph && (!has_moe || fused) && !logits_out && !tp;
    const int attn_ctx = use_graph ? c.max_ctx : max_ctx;     // LDS sizing of the attention kernel must not depend on the step
    if (use_graph) {
        auto it = m->batch_graphs.find(n);
        if (it != m->batch_graphs.end()) {
            IFA_HIP_CHECK(hipGraphLaunch(it->second, m->stream));
            IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
            if ((rc = wait_err_check("batched decode step"))) { drop_graphs(m); return rc; }      // (the captured steps hold K-parts launches: re-captured without them)
            if (next_tokens) for (int r = 0; r < n; r++) next_tokens[r] = m->host_pinned[8 + r];
            return IFA_OK;
        }
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
        IFA_HIP_CHECK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    }
    auto body = [&]() -> int {
    IFA_HIP_CHECK(hipMemcpyAsync(m->batch_tab_dev, m->batch_tab_pin, tab_bytes, hipMemcpyHostToDevice, m->stream));
    if (fused)
        k_dec_batch_gather<<<dim3(4, (unsigned)T), dim3(256), 0, m->stream>>>((const half_t *)m->g[T_EMBD].data, tok_d, pos_d, (int)D, (int)m->g[T_EMBD].rows,
                                                                              m->x, c.rope_order ? m->brope : nullptr, c.head_dim, c.rope_theta,
                                                                              (int)(c.head_dim * c.partial_rotary + 0.5f), c.embd_scale);
    else
        k_gather_rows<<<dim3(4, (unsigned)T), dim3(256), 0, m->stream>>>((const half_t *)m->g[T_EMBD].data, tok_d, T, (int)D,
                                                                          (int)m->g[T_EMBD].rows, m->x, c.embd_scale);
    IFA_LAUNCH_CHECK();
    half_t *x = m->x;
    const Tensor none;
    const bool seq_wiring = !c.parallel_attn && !c.share_input;
    bool xn_ready = false;           // see forward_ops: every residual Add is fused with the norm that follows it
    if (fused) {
        for (int l = 0; l < c.layers; l++) {
            if ((rc = batch_fused_layer(m, l, n, x, m->f, rows_d + (size_t)l * (size_t)n))) return rc;
            std::swap(m->x, m->f);
            x = m->x;
        }
    }
    for (int l = fused ? c.layers : 0; l < c.layers; l++) {
        Layer &L = m->layers[(size_t)l];
        const half_t *attn_in = x;
        if (L.t[T_ATTN_NORM].present()) {
            if (!xn_ready && (rc = norm_rows(m, x, T, L.t[T_ATTN_NORM], L.t[T_ATTN_NORM_B], m->xn, c.attn_norm_base))) return rc;
            attn_in = m->xn;
        }
        xn_ready = false;
        if ((rc = matmul(m, attn_in, T, L.t[T_WQ], L.t[T_WQ_B], m->q))) return rc;
        if ((rc = matmul(m, attn_in, T, L.t[T_WK], L.t[T_WK_B], m->k))) return rc;
        if ((rc = matmul(m, attn_in, T, L.t[T_WV], L.t[T_WV_B], m->v))) return rc;
        if (c.rope_order != 0) {
            if ((rc = ifa_rope_rows(m->q, c.head_dim, c.heads, T, pos_d, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
            if ((rc = ifa_rope_rows(m->k, c.head_dim, c.kv_heads, T, pos_d, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
        }
        const AttnRowH *lr = rows_d + (size_t)l * (size_t)n;
        if (c.kv_dtype == Q8_B32T2) k_kv_store_rows<true><<<dim3((unsigned)n, 2), dim3(256), 0, m->stream>>>(m->k, m->v, (int)KVD, m->kv_row_bytes, lr);
        else k_kv_store_rows<false><<<dim3((unsigned)n, 2), dim3(256), 0, m->stream>>>(m->k, m->v, (int)KVD, m->kv_row_bytes, lr);
        IFA_LAUNCH_CHECK();
        if ((rc = ifa_attention_rows(m->q, lr, c.kv_dtype, n, attn_ctx, c.heads, c.kv_heads, c.head_dim, c.use_alibi ? 1.0f : c.kq_scale,
                                     c.use_alibi, c.tp_rank * c.heads, c.heads * std::max(1, c.tp_size), m->att, s))) return rc;
        if ((rc = matmul(m, m->att, T, L.t[T_WO], merging ? none : L.t[T_WO_B], m->a))) return rc;
        if ((rc = tp_merge_rows(m, m->a, T, L.t[T_WO_B]))) return rc;
        if ((rc = layer_tail_ops(m, l, T, x, attn_in, xn_ready))) return rc;
    }
    if (scale_on(c.out_scale) && (rc = ifa_scale(x, c.out_scale, (size_t)T * D, x, s))) return rc;
    const half_t *hfin = x;
    if (m->g[T_OUT_NORM].present()) {
        if (!xn_ready && (rc = norm_rows(m, x, T, m->g[T_OUT_NORM], m->g[T_OUT_NORM_B], m->xn, c.out_norm_base))) return rc;
        hfin = m->xn;
    }
    const Tensor &lm = m->g[T_LM_HEAD];
    const size_t V = lm.rows;
    if ((rc = matmul(m, hfin, T, lm, none, m->logits))) return rc;
    if (logits_out) IFA_HIP_CHECK(hipMemcpyAsync(logits_out, m->logits, (size_t)T * V * 2, hipMemcpyDeviceToDevice, m->stream));
    if (tp) {                // one distributed argmax per row over the group's vocabulary shards
        if ((rc = tp_pick_rows(m, *tp, m->logits, V, (int)V, n))) return rc;
        IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 8, m->tp_tok, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, m->stream));
        return IFA_OK;
    }
    if ((rc = ifa_argmax_rows(m->logits, V, V, (size_t)n, m->state + 8, m->state + 3, s))) return rc;      // one launch for the n rows
    IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 8, m->state + 8, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, m->stream));
    return IFA_OK;
    };
    rc = body();
    if (use_graph) {
        // forward ops swap m->x / m->f per layer: an odd layer count would leave them exchanged between replays
        hipGraph_t gph = nullptr;
        hipError_t e = hipStreamEndCapture(m->stream, &gph);
        if (rc) { if (gph) (void)hipGraphDestroy(gph); return rc; }
        if (e != hipSuccess) return ifa_fail(IFA_ERR_HIP, "hipStreamEndCapture (batched step): %s", hipGetErrorString(e));
        hipGraphExec_t ex = nullptr;
        IFA_HIP_CHECK(hipGraphInstantiate(&ex, gph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(gph);
        m->batch_graphs[n] = ex;
        IFA_HIP_CHECK(hipGraphLaunch(ex, m->stream));
    } else if (rc) return rc;
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    if ((rc = wait_err_check("batched decode step"))) { drop_graphs(m); return rc; }
    if (next_tokens) for (int r = 0; r < n; r++) next_tokens[r] = m->host_pinned[8 + r];
    return IFA_OK;
}

extern "C" {

int ifa_model_create(const ifa_model_config *cfg, ifa_model **out)
{
    IFA_REQUIRE(cfg && out, "ifa_model_create: null pointer");
    IFA_REQUIRE(cfg->dim > 0 && cfg->layers > 0 && cfg->heads > 0 && cfg->kv_heads > 0 && cfg->head_dim > 0
                && cfg->vocab > 0 && cfg->max_ctx > 0, "ifa_model_create: bad hyper-parameters");
    IFA_REQUIRE(cfg->heads % cfg->kv_heads == 0, "ifa_model_create: heads %d not a multiple of kv_heads %d", cfg->heads, cfg->kv_heads);
    IFA_REQUIRE(cfg->kv_dtype == F16 || cfg->kv_dtype == Q8_B32T2, "ifa_model_create: kv dtype %d", cfg->kv_dtype);
    IFA_REQUIRE(cfg->norm_kind == 0 || cfg->norm_kind == 1, "ifa_model_create: norm_kind %d", cfg->norm_kind);
    IFA_HIP_CHECK(hipSetDevice(cfg->device));
    ifa_model *m = new ifa_model();
    m->cfg = *cfg;
    if (m->cfg.eps <= 0) m->cfg.eps = 1e-5f;
    if (m->cfg.kq_scale <= 0) m->cfg.kq_scale = 1.0f;
    if (m->cfg.partial_rotary <= 0) m->cfg.partial_rotary = 1.0f;
    if (m->cfg.tp_size <= 0) m->cfg.tp_size = 1;
    for (float *sc : {&m->cfg.attn_out_scale, &m->cfg.ffn_out_scale, &m->cfg.out_scale}) if (*sc <= 0.0f) *sc = 1.0f;
    if (m->cfg.embd_scale < 0.0f) m->cfg.embd_scale = sqrtf((float)m->cfg.dim);      // LinearNorm's default scale (tensor_opr.cu:492-494)
    m->layers.resize((size_t)cfg->layers);
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete m; return ifa_fail(IFA_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    *out = m;
    return IFA_OK;
}

int ifa_model_destroy(ifa_model *m)
{
    if (!m) return IFA_OK;
    (void)hipSetDevice(m->cfg.device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    drop_graphs(m);
    for (auto &sl : m->slots) {
        for (void *p : sl.k) if (p) (void)hipFree(p);
        for (void *p : sl.v) if (p) (void)hipFree(p);
    }
    for (Layer &L : m->layers) {
        for (Tensor &t : L.t) free_tensor(t);
        for (Tensor &t : L.experts) free_tensor(t);
        if (L.moe_table) (void)hipFree(L.moe_table);
        if (L.moe_table_aos) (void)hipFree(L.moe_table_aos);
        if (L.moe_table_mo) (void)hipFree(L.moe_table_mo);
        if (L.kcache) (void)hipFree(L.kcache);
        if (L.vcache) (void)hipFree(L.vcache);
    }
    if (m->batch_tab_dev) (void)hipFree(m->batch_tab_dev);
    if (m->batch_tab_pin) (void)hipHostFree(m->batch_tab_pin);
    if (m->attn_ws.S) (void)hipFree(m->attn_ws.S);
    if (m->attn_ws.lmax) (void)hipFree(m->attn_ws.lmax);
    if (m->attn_ws.opart) (void)hipFree(m->attn_ws.opart);
    for (Tensor &t : m->g) free_tensor(t);
    half_t **bufs[] = {&m->x, &m->x2, &m->xn, &m->hn, &m->pn, &m->q, &m->k, &m->v, &m->dqkv, &m->bqkv, &m->att, &m->a, &m->f, &m->t1, &m->t2, &m->logits,
                       &m->moe_gate, &m->moe_out, &m->moe_in, &m->moe_wdev};
    for (half_t **b : bufs) if (*b) (void)hipFree(*b);
    if (m->moe_idx) (void)hipFree(m->moe_idx);
    if (m->moe_route) (void)hipFree(m->moe_route);
    if (m->moe_pin) (void)hipHostFree(m->moe_pin);
    if (m->trace) (void)hipFree(m->trace);
    if (m->xq) (void)hipFree(m->xq);
    if (m->attq) (void)hipFree(m->attq);
    { void *mb[] = {m->moe_sel, m->moe_epos, m->moe_counts, m->moe_selw, m->moe_g1, m->moe_g3, m->moe_gin, m->moe_gout, m->moe_xq_in,
                    m->moe_xq_mid, m->moe_tiles, m->moe_singles, m->moe_smalls};
      for (void *b : mb) if (b) (void)hipFree(b); }
    { void *tpb[] = {m->tp_a, m->tp_f, m->tp_hid, m->tp_logits, m->tp_best, m->tp_gather, m->tp_tok};
      for (void *b : tpb) if (b) (void)hipFree(b); }
    if (m->state) (void)hipFree(m->state);
    if (m->rope_tab) (void)hipFree(m->rope_tab);
    if (m->tokens_dev) (void)hipFree(m->tokens_dev);
    if (m->host_pinned) (void)hipHostFree(m->host_pinned);
    if (m->qa_gran) (void)hipFree(m->qa_gran);
    if (m->ch_gran) (void)hipFree(m->ch_gran);
    if (m->ch_flags) (void)hipFree(m->ch_flags);
    if (m->qa_call) (void)hipFree(m->qa_call);
    if (m->qa_err) (void)hipFree(m->qa_err);
    if (m->st_keys) (void)hipFree(m->st_keys);
    if (m->st_counter) (void)hipFree(m->st_counter);
    if (m->stream) (void)ifa_gemm_release_stream((ifa_stream)m->stream);
    if (m->stream && m->own_stream) (void)hipStreamDestroy(m->stream);
    if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    delete m;
    return IFA_OK;
}

int ifa_model_set_tensor(ifa_model *m, int layer, int tensor_id, int expert, int dtype, const void *dev_src,
                         size_t rows, size_t cols)
{
    IFA_REQUIRE(m && dev_src, "ifa_model_set_tensor: null pointer");
    IFA_REQUIRE(tensor_id >= 0 && tensor_id < T_MAX, "ifa_model_set_tensor: tensor id %d", tensor_id);
    IFA_REQUIRE(expert < 0 || (expert < m->cfg.experts && layer >= 0 && (tensor_id == T_W1 || tensor_id == T_W2 || tensor_id == T_W3)),
                "ifa_model_set_tensor: expert %d (of %d) / tensor %d", expert, m->cfg.experts, tensor_id);
    IFA_REQUIRE(block_capacity(dtype) > 0 && dtype != F32, "ifa_model_set_tensor: dtype %d", dtype);
    IFA_REQUIRE(cols % (size_t)block_capacity(dtype) == 0, "ifa_model_set_tensor: cols %zu vs block capacity", cols);
    {   // the scratch buffers are sized from the config and the kernels are launched with the tensor's dimensions: a
        // mismatch would write out of bounds on the device, so it is refused here (per-shard dimensions under TP)
        const ifa_model_config &c = m->cfg;
        const size_t D = (size_t)c.dim, QD = (size_t)c.heads * c.head_dim, KVD = (size_t)c.kv_heads * c.head_dim, F = (size_t)c.ffn;
        size_t er = 0, ec = 0;        // expected rows / cols; 0 = free
        bool vec = false;             // [1][n] vectors (norm weights, biases) may also arrive as [n][1]
        switch (tensor_id) {
        case T_EMBD: ec = D; break;
        case T_LM_HEAD: ec = D; break;                      // rows: the vocabulary (or this rank's shard of it)
        case T_OUT_NORM: case T_OUT_NORM_B: case T_ATTN_NORM: case T_ATTN_NORM_B: case T_FFN_NORM: case T_FFN_NORM_B:
        case T_WO_B: case T_W2_B: vec = true; ec = D; break;
        case T_WQ: er = QD; ec = D; break;
        case T_WK: case T_WV: er = KVD; ec = D; break;
        case T_WO: er = D; ec = QD; break;
        case T_W1: case T_W3: er = F; ec = D; break;
        case T_W2: er = D; ec = F; break;
        case T_MOE_GATE: er = (size_t)c.experts; ec = D; break;
        case T_WQ_B: vec = true; ec = QD; break;
        case T_WK_B: case T_WV_B: vec = true; ec = KVD; break;
        case T_W1_B: case T_W3_B: vec = true; ec = F; break;
        default: break;
        }
        if (vec) IFA_REQUIRE(rows * cols == ec, "ifa_model_set_tensor: tensor %d holds %zu x %zu values, the model needs %zu", tensor_id, rows, cols, ec);
        else {
            IFA_REQUIRE(ec == 0 || cols == ec, "ifa_model_set_tensor: tensor %d has %zu columns, the model needs %zu", tensor_id, cols, ec);
            IFA_REQUIRE(er == 0 || rows == er, "ifa_model_set_tensor: tensor %d has %zu rows, the model needs %zu", tensor_id, rows, er);
        }
        if (tensor_id == T_LM_HEAD && layer < 0) IFA_REQUIRE(rows <= (size_t)c.vocab, "ifa_model_set_tensor: lm_head has %zu rows, vocab is %d", rows, c.vocab);
        if (tensor_id == T_EMBD) IFA_REQUIRE(rows <= (size_t)c.vocab, "ifa_model_set_tensor: %zu embedding rows, vocab is %d", rows, c.vocab);
    }
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    Tensor *t;
    if (tensor_id < 10) t = &m->g[tensor_id];
    else {
        IFA_REQUIRE(layer >= 0 && layer < m->cfg.layers, "ifa_model_set_tensor: layer %d", layer);
        Layer &L = m->layers[(size_t)layer];
        if (expert >= 0) {
            if (L.experts.empty()) L.experts.resize((size_t)m->cfg.experts * 3);
            t = &L.experts[(size_t)expert * 3 + (tensor_id == T_W1 ? 0 : (tensor_id == T_W2 ? 1 : 2))];
        } else t = &L.t[tensor_id];
    }
    free_tensor(*t);
    if (expert >= 0) {      // the layer's table of MO copies points into the copy just freed: ensure_mo rebuilds it
        Layer &Lx = m->layers[(size_t)layer];
        if (Lx.moe_table_mo) { (void)hipFree(Lx.moe_table_mo); Lx.moe_table_mo = nullptr; }
    }
    // load time: the source may have been produced on another stream (e.g. the caller's default stream)
    IFA_HIP_CHECK(hipDeviceSynchronize());
    const size_t bytes = rows * ifa_row_bytes(dtype, cols);
    IFA_HIP_CHECK(hipMalloc(&t->data, bytes));
    IFA_HIP_CHECK(hipMemcpyAsync(t->data, dev_src, bytes, hipMemcpyDeviceToDevice, m->stream));
    t->dtype = dtype; t->rows = rows; t->cols = cols;
    const bool is_matrix = (layer < 0 && tensor_id == T_LM_HEAD) || tensor_id == T_WQ || tensor_id == T_WK || tensor_id == T_WV || tensor_id == T_WO
        || tensor_id == T_W1 || tensor_id == T_W2 || tensor_id == T_W3;
    if (is_matrix && ax8_eligible(dtype)) {
        IFA_HIP_CHECK(hipMalloc(&t->tiled, rows * ifa_tiled_row_bytes(dtype, cols)));
        int rc = ifa_repack_weights(dtype, t->data, rows, cols, t->tiled, m->stream);
        if (rc) return rc;
    }
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    drop_graphs(m);
    return IFA_OK;
}

int ifa_model_set_tensor_f16(ifa_model *m, int layer, int tensor_id, int expert, int target_dtype,
                             const void *dev_src_f16, size_t rows, size_t cols)
{
    IFA_REQUIRE(m && dev_src_f16, "ifa_model_set_tensor_f16: null pointer");
    if (target_dtype == F16) return ifa_model_set_tensor(m, layer, tensor_id, expert, F16, dev_src_f16, rows, cols);
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    void *tmp = nullptr;
    const size_t bytes = rows * ifa_row_bytes(target_dtype, cols);
    IFA_REQUIRE(bytes > 0, "ifa_model_set_tensor_f16: dtype %d", target_dtype);
    IFA_HIP_CHECK(hipMalloc(&tmp, bytes));
    IFA_HIP_CHECK(hipDeviceSynchronize());   // source may come from another stream
    int rc = ifa_quantize(target_dtype, dev_src_f16, rows, cols, tmp, m->stream);   // DeviceTensorBuilder::Build_Quant
    if (rc == IFA_OK) rc = ifa_model_set_tensor(m, layer, tensor_id, expert, target_dtype, tmp, rows, cols);
    (void)hipStreamSynchronize(m->stream);
    (void)hipFree(tmp);
    return rc;
}

int ifa_model_finalize(ifa_model *m)
{
    IFA_REQUIRE(m, "ifa_model_finalize: null model");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    (void)wait_err_word();       // (the device's wait-error word exists before any step can be captured: ifa_host.h)
    const ifa_model_config &c = m->cfg;
    const size_t KVD = (size_t)c.kv_heads * c.head_dim;
    IFA_REQUIRE(c.kv_dtype != Q8_B32T2 || KVD % 32 == 0, "Q8 KV cache needs kv_dim %% 32 == 0");
    m->kv_row_bytes = ifa_row_bytes(c.kv_dtype, KVD);
    for (Layer &L : m->layers) {
        if (!L.kcache) {   // KVCache::Init (kv_cache.cc:278-319): (kv_dim, max_ctx) per layer
            // (at least DEC_ATTN_MIN_ROWS rows are allocated: the fused attention kernel requests its first 256 rows unclamped)
            const size_t kv_alloc = m->kv_row_bytes * (size_t)std::max(c.max_ctx, DEC_ATTN_MIN_ROWS);
            IFA_HIP_CHECK(hipMalloc(&L.kcache, kv_alloc));
            IFA_HIP_CHECK(hipMalloc(&L.vcache, kv_alloc));
            IFA_HIP_CHECK(hipMemsetAsync(L.kcache, 0, kv_alloc, m->stream));
            IFA_HIP_CHECK(hipMemsetAsync(L.vcache, 0, kv_alloc, m->stream));
        }
        if (c.experts > 0 && (int)L.experts.size() == c.experts * 3) {      // pointer table for the fused MoE kernels
            std::vector<void *> tab((size_t)c.experts * 4, nullptr);
            for (int e = 0; e < c.experts; e++) {
                tab[(size_t)e * 4 + 0] = L.experts[(size_t)e * 3 + 0].tiled;
                tab[(size_t)e * 4 + 1] = L.experts[(size_t)e * 3 + 2].tiled;
                tab[(size_t)e * 4 + 2] = L.experts[(size_t)e * 3 + 1].tiled;
            }
            if (!L.moe_table) IFA_HIP_CHECK(hipMalloc(&L.moe_table, tab.size() * sizeof(void *)));
            IFA_HIP_CHECK(hipMemcpy(L.moe_table, tab.data(), tab.size() * sizeof(void *), hipMemcpyHostToDevice));
            std::vector<void *> aos((size_t)c.experts * 3, nullptr);
            for (int e = 0; e < c.experts; e++) for (int k3 = 0; k3 < 3; k3++) aos[(size_t)e * 3 + k3] = L.experts[(size_t)e * 3 + k3].data;
            if (!L.moe_table_aos) IFA_HIP_CHECK(hipMalloc(&L.moe_table_aos, aos.size() * sizeof(void *)));
            IFA_HIP_CHECK(hipMemcpy(L.moe_table_aos, aos.data(), aos.size() * sizeof(void *), hipMemcpyHostToDevice));
        }
    }
    if (!m->attn_ws.S) {
        IFA_HIP_CHECK(hipMalloc((void **)&m->attn_ws.S, (size_t)c.heads * c.max_ctx * 2));
        IFA_HIP_CHECK(hipMalloc((void **)&m->attn_ws.lmax, (size_t)c.heads * DEC_ATTN_MAX_SPLITS * 4));
        IFA_HIP_CHECK(hipMalloc((void **)&m->attn_ws.opart, (size_t)c.heads * DEC_ATTN_MAX_SPLITS * c.head_dim * 4));
    }
    if (!m->state) {
        IFA_HIP_CHECK(hipMalloc((void **)&m->state, sizeof(int) * (8 + ifa_model::RING)));
        IFA_HIP_CHECK(hipMemsetAsync(m->state, 0, sizeof(int) * (8 + ifa_model::RING), m->stream));
        IFA_HIP_CHECK(hipMalloc((void **)&m->rope_tab, sizeof(float) * (size_t)c.head_dim));
        IFA_HIP_CHECK(hipHostMalloc((void **)&m->host_pinned, sizeof(int) * (16 + ifa_model::RING), hipHostMallocDefault));
    }
    int rc = ensure_scratch(m, 1);
    if (rc) return rc;
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    m->finalized = true;
    return IFA_OK;
}

int ifa_model_reset(ifa_model *m)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_reset: model not finalized");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    for (Layer &L : m->layers) {
        IFA_HIP_CHECK(hipMemsetAsync(L.kcache, 0, m->kv_row_bytes * (size_t)m->cfg.max_ctx, m->stream));
        IFA_HIP_CHECK(hipMemsetAsync(L.vcache, 0, m->kv_row_bytes * (size_t)m->cfg.max_ctx, m->stream));
    }
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    return IFA_OK;
}

int ifa_model_kv_slots(ifa_model *m, int n_slots)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_kv_slots: model not finalized");
    IFA_REQUIRE(n_slots >= 1 && n_slots <= 4096, "ifa_model_kv_slots: n_slots %d", n_slots);
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    if (m->slots.empty()) m->slots.resize(1);        // slot 0 = the cache finalize() made (active, nothing parked)
    IFA_REQUIRE((int)m->slots.size() <= n_slots, "ifa_model_kv_slots: cannot shrink below %zu slots", m->slots.size());
    const size_t bytes = m->kv_row_bytes * (size_t)std::max(m->cfg.max_ctx, DEC_ATTN_MIN_ROWS);
    while ((int)m->slots.size() < n_slots) {
        ifa_model::KvSlot sl;
        for (size_t l = 0; l < m->layers.size(); l++) {
            void *k = nullptr, *v = nullptr;
            IFA_HIP_CHECK(hipMalloc(&k, bytes));
            IFA_HIP_CHECK(hipMalloc(&v, bytes));
            IFA_HIP_CHECK(hipMemsetAsync(k, 0, bytes, m->stream));
            IFA_HIP_CHECK(hipMemsetAsync(v, 0, bytes, m->stream));
            sl.k.push_back(k); sl.v.push_back(v);
        }
        m->slots.push_back(std::move(sl));
    }
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    return IFA_OK;
}

int ifa_model_select_kv(ifa_model *m, int slot)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_select_kv: model not finalized");
    if (m->slots.empty()) m->slots.resize(1);
    IFA_REQUIRE(slot >= 0 && slot < (int)m->slots.size(), "ifa_model_select_kv: slot %d of %zu", slot, m->slots.size());
    if (slot == m->cur_slot) return IFA_OK;
    // the captured multi-GPU step holds the outgoing slot's cache pointers
    if (m->tp_graph_exec) { (void)hipGraphExecDestroy(m->tp_graph_exec); m->tp_graph_exec = nullptr; }
    if (m->tp_graph) { (void)hipGraphDestroy(m->tp_graph); m->tp_graph = nullptr; }
    ifa_model::KvSlot &out = m->slots[(size_t)m->cur_slot], &in = m->slots[(size_t)slot];
    out.k.clear(); out.v.clear();
    for (Layer &L : m->layers) { out.k.push_back(L.kcache); out.v.push_back(L.vcache); }
    out.graph = m->graph; out.exec = m->graph_exec;
    if (m->graph_exec_n) { (void)hipGraphExecDestroy(m->graph_exec_n); m->graph_exec_n = nullptr; }      // (the multi-step replay is not kept per slot)
    if (m->graph_n) { (void)hipGraphDestroy(m->graph_n); m->graph_n = nullptr; }
    for (size_t l = 0; l < m->layers.size(); l++) { m->layers[l].kcache = in.k[l]; m->layers[l].vcache = in.v[l]; }
    m->graph = in.graph; m->graph_exec = in.exec;
    in.k.clear(); in.v.clear(); in.graph = nullptr; in.exec = nullptr;
    m->cur_slot = slot;
    return IFA_OK;
}

int ifa_model_set_option(ifa_model *m, const char *name, int value)
{
    IFA_REQUIRE(m && name, "ifa_model_set_option: null pointer");
    struct { const char *n; int *p; } opts[] = {
        {"fused", &m->opt_fused}, {"graph", &m->opt_graph}, {"rpw_qkv", &m->opt_rpw_qkv}, {"rpw_wo", &m->opt_rpw_wo},
        {"rpw_ffn", &m->opt_rpw_ffn}, {"rpw_w2", &m->opt_rpw_w2}, {"rpw_lm", &m->opt_rpw_lm}, {"trace", &m->opt_trace},
        {"bench_mode", &m->opt_bench_mode}, {"touch_stride", &m->opt_touch_stride}, {"attn_split_ctx", &m->opt_attn_split_ctx}, {"batch_graph", &m->opt_batch_graph}, {"gemm_rows", &m->opt_gemm_rows}, {"batch_fused", &m->opt_batch_fused}, {"rows_mo", &m->opt_rows_mo}, {"debug_mo_alloc_fail", &m->opt_debug_mo_alloc_fail}, {"rows_kparts", &m->opt_rows_kparts}, {"prefill_chunk", &m->opt_prefill_chunk}, {"prefill_big_min", &m->opt_prefill_big_min}, {"prefill_mid", &m->opt_prefill_mid}, {"prefill_mid_max", &m->opt_prefill_mid_max}, {"gemm_splitk", &m->opt_gemm_splitk}, {"moe_singles", &m->opt_moe_singles}, {"moe_overlap", &m->opt_moe_overlap}, {"prefill_big", &m->opt_prefill_big}, {"moe_router_fused", &m->opt_moe_router_fused}, {"tp_fuse_add", &m->opt_tp_fuse_add}, {"attn_q8", &m->opt_attn_q8}, {"attn_kt", &m->opt_attn_kt}, {"fuse_attn", &m->opt_fuse_attn}, {"fuse_ffn", &m->opt_fuse_ffn}, {"attn_post_as_residual", &m->opt_attn_post_as_residual}, {"chain_late_w2", &m->opt_chain_late_w2}, {"fuse_attn_timeout_us", &m->opt_fuse_attn_timeout_us}, {"step_tail", &m->opt_step_tail}, {"graph_steps", &m->opt_graph_steps}, {"moe_device", &m->opt_moe_device}, 
        
        {"debug_layers", &m->opt_debug_layers}, {"debug_layer0", &m->opt_debug_layer0}, {"debug_hidden_in", &m->opt_debug_hidden_in}};
    for (auto &o : opts)
        if (strcmp(o.n, name) == 0) {
            *o.p = value;
            drop_graphs(m);
            return IFA_OK;
        }
    return ifa_fail(IFA_ERR_ARG, "ifa_model_set_option: unknown option '%s'", name);
}

// ids the greedy selection never offers (GetSortedTopK skips the vocabulary's unk id and Invalid-type tokens,
// src/transformer/sampling_strategy.cc:281-297): kept in the device state next to token / position so that the
// captured step needs no re-capture when they change
int ifa_model_set_excluded_tokens(ifa_model *m, const int *ids_host, int n)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_set_excluded_tokens: model not finalized");
    IFA_REQUIRE(n >= 0 && n <= 3 && (n == 0 || ids_host), "ifa_model_set_excluded_tokens: n %d (0..3)", n);
    int v[4] = {n, -1, -1, -1};
    for (int i = 0; i < n; i++) v[1 + i] = ids_host[i];
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    IFA_HIP_CHECK(hipMemcpy(m->state + 3, v, sizeof(v), hipMemcpyHostToDevice));
    return IFA_OK;
}

int ifa_model_fused_supported(ifa_model *m, char *why, size_t why_len)
{
    IFA_REQUIRE(m, "ifa_model_fused_supported: null model");
    std::string w;
    bool ok = fused_supported(m, &w);
    if (why && why_len) { strncpy(why, w.c_str(), why_len - 1); why[why_len - 1] = 0; }
    return ok ? 1 : 0;
}

int ifa_model_forward(ifa_model *m, const int *tokens_host, int n_tokens, int prefix_len, void *logits_out_dev,
                      int *next_token_host)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_forward: model not finalized");
    IFA_REQUIRE(tokens_host, "ifa_model_forward: null tokens");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    // Round 5: prompts of 34..48 tokens as TWO passes of the rows GEMM (32 tokens, then 2..16: the weights stream into registers five
    // groups deep) instead of one pass of the op-by-op layer: 40 tokens 7.08 -> 6.35 ms, 48 tokens 7.26 -> 6.67 (profiles/r05_prompt_lengths.log;
    // two passes of 17..32 rows each -- 49..64 tokens -- measured no faster than the tile kernels).  The second pass reads the first
    // one's K / V rows from the cache like any continued prompt; every row goes through the kernels of a prompt of <= 32 tokens.  Never
    // a one-token pass: a single row takes the int8 GEMV (the reference's rule for ONE row), which is not how a prompt's rows are computed.
    if (m->opt_prefill_chunk && !m->topo && n_tokens >= 34 && n_tokens <= 48 && m->cfg.experts == 0 && batch_fused_ok(m, 32)
        && prefix_len >= 0 && prefix_len + n_tokens <= m->cfg.max_ctx) {
        const int t1 = 32;
        const size_t V = m->g[T_LM_HEAD].rows;
        int rc = forward_ops(m, tokens_host, t1, prefix_len, logits_out_dev, nullptr, true);
        if (rc) return rc;
        return forward_ops(m, tokens_host + t1, n_tokens - t1, prefix_len + t1, logits_out_dev ? (char *)logits_out_dev + (size_t)t1 * V * 2 : nullptr, next_token_host);
    }
    return forward_ops(m, tokens_host, n_tokens, prefix_len, logits_out_dev, next_token_host);
}

static int decode_impl(ifa_model *m, int first_token, int start_pos, int n_steps, int *out_tokens_host, float *elapsed_ms, bool prepare_only);
int ifa_model_decode(ifa_model *m, int first_token, int start_pos, int n_steps, int *out_tokens_host, float *elapsed_ms)
{
    return decode_impl(m, first_token, start_pos, n_steps, out_tokens_host, elapsed_ms, false);
}

// Everything a decode call of n_steps from start_pos sets up before its first launch -- the attention variant of the contexts it
// reaches, the hand-off arenas, the captured step(s) -- without running a step: a caller that times its first call (bench.py with
// --warmup 0, a service's first request) keeps graph capture / instantiation out of it.  The KV cache and the activations are not
// touched (the token / position words are rewritten by every call anyway).
int ifa_model_decode_prepare(ifa_model *m, int start_pos, int n_steps)
{
    return decode_impl(m, 0, start_pos, n_steps, nullptr, nullptr, true);
}

static int decode_impl(ifa_model *m, int first_token, int start_pos, int n_steps, int *out_tokens_host, float *elapsed_ms, bool prepare_only)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_decode: model not finalized");
    IFA_REQUIRE(n_steps > 0 && n_steps <= ifa_model::RING, "ifa_model_decode: n_steps %d (max %d per call)", n_steps, ifa_model::RING);
    IFA_REQUIRE(start_pos >= 0 && start_pos + n_steps <= m->cfg.max_ctx, "ifa_model_decode: positions [%d,%d) exceed max_ctx %d",
                start_pos, start_pos + n_steps, m->cfg.max_ctx);
    IFA_REQUIRE(m->g[T_EMBD].present() && m->g[T_LM_HEAD].present(), "ifa_model_decode: embeddings / lm_head missing (pipeline stage worker)");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    std::string why;
    if (!m->opt_fused || !fused_supported(m, &why)) {
        // op-by-op fallback: same semantics, host-driven
        if (prepare_only) return IFA_OK;
        int tok = first_token;
        for (int i = 0; i < n_steps; i++) {
            int nt = 0;
            int rc = forward_ops(m, &tok, 1, start_pos + i, nullptr, &nt);
            if (rc) return rc;
            if (out_tokens_host) out_tokens_host[i] = nt;
            tok = nt;
        }
        if (elapsed_ms) *elapsed_ms = -1.0f;
        return IFA_OK;
    }
    static const bool trace_host = getenv("IFA_TRACE_DECODE") != nullptr;       // tuning aid: host-side timeline of the call on stderr
    const auto th0 = std::chrono::steady_clock::now();
    auto th = [&](const char *what) {
        if (trace_host) fprintf(stderr, "decode-host %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - th0).count());
    };
    int rc = ensure_scratch(m, 1);
    if (rc) return rc;
    hipStream_t s = m->stream;
    // attention variant of this call: one workgroup per head, or keys split over workgroups once the context the
    // call reaches passes the threshold (the captured step is re-captured when the variant changes)
    choose_attn_split(m, start_pos + n_steps);
    if ((rc = qkv_attn_ready(m))) return rc;
    if ((rc = step_tail_ready(m))) return rc;
    m->host_pinned[0] = first_token; m->host_pinned[1] = start_pos; m->host_pinned[2] = 0;
    IFA_HIP_CHECK(hipMemcpyAsync(m->state, m->host_pinned, 3 * sizeof(int), hipMemcpyHostToDevice, s));
    if (m->qa_on || m->ch_on) {      // the granule tags of this call: (call counter, position) -- consecutive steps never share one
        m->qa_calls = (m->qa_calls % 4000u) + 1u;
        m->host_pinned[6] = (int)m->qa_calls;
        IFA_HIP_CHECK(hipMemcpyAsync(m->qa_call, m->host_pinned + 6, sizeof(int), hipMemcpyHostToDevice, s));
    }
    if (m->opt_graph && !m->graph_exec) {
        IFA_HIP_CHECK(hipStreamSynchronize(s));
        IFA_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        rc = enqueue_fused_step(m);
        hipGraph_t gph = nullptr;
        hipError_t e = hipStreamEndCapture(s, &gph);
        if (rc) { if (gph) (void)hipGraphDestroy(gph); return rc; }
        if (e != hipSuccess) return ifa_fail(IFA_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        if (m->graph) (void)hipGraphDestroy(m->graph);
        m->graph = gph;
        IFA_HIP_CHECK(hipGraphInstantiate(&m->graph_exec, m->graph, nullptr, nullptr, 0));
    }
    const int S = m->opt_graph_steps;
    if (m->opt_graph && S > 1 && n_steps >= S && (!m->graph_exec_n || m->graph_n_steps != S)) {
        if (m->graph_exec_n) { (void)hipGraphExecDestroy(m->graph_exec_n); m->graph_exec_n = nullptr; }
        if (m->graph_n) { (void)hipGraphDestroy(m->graph_n); m->graph_n = nullptr; }
        IFA_HIP_CHECK(hipStreamSynchronize(s));
        IFA_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < S && !rc; i++) rc = enqueue_fused_step(m);
        hipGraph_t gph = nullptr;
        hipError_t e = hipStreamEndCapture(s, &gph);
        if (rc) { if (gph) (void)hipGraphDestroy(gph); return rc; }
        if (e != hipSuccess) return ifa_fail(IFA_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        m->graph_n = gph; m->graph_n_steps = S;
        IFA_HIP_CHECK(hipGraphInstantiate(&m->graph_exec_n, m->graph_n, nullptr, nullptr, 0));
    }
    if (prepare_only) return IFA_OK;
    th("state copies enqueued");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (elapsed_ms) { IFA_HIP_CHECK(hipEventCreate(&e0)); IFA_HIP_CHECK(hipEventCreate(&e1)); IFA_HIP_CHECK(hipEventRecord(e0, s)); }
    th("events created, first recorded");
    if (m->st_on && (rc = launch_gather(m))) return rc;      // the call's first step: its input is gathered here, every later one by the step before it
    for (int i = 0; i < n_steps;) {
        if (m->opt_graph && m->graph_exec_n && m->graph_n_steps == S && S > 1 && n_steps - i >= S) { IFA_HIP_CHECK(hipGraphLaunch(m->graph_exec_n, s)); i += S; continue; }
        if (m->opt_graph) IFA_HIP_CHECK(hipGraphLaunch(m->graph_exec, s));
        else if ((rc = enqueue_fused_step(m))) return rc;
        i++;
    }
    th("steps enqueued");
    if (elapsed_ms) IFA_HIP_CHECK(hipEventRecord(e1, s));
    IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 8, m->state + 8, sizeof(int) * (size_t)n_steps, hipMemcpyDeviceToHost, s));
    int *qerr = m->host_pinned + 8 + ifa_model::RING;
    qerr[0] = 0;
    if (m->qa_on || m->ch_on) IFA_HIP_CHECK(hipMemcpyAsync(qerr, m->qa_err, 4, hipMemcpyDeviceToHost, s));
    th("copies back enqueued");
    IFA_HIP_CHECK(hipStreamSynchronize(s));
    th("stream synchronised");
    if (qerr[0] != 0) {      // a head's workgroup gave up waiting for its q | k | v rows: the step's results are not valid
        (void)hipMemsetAsync(m->qa_err, 0, 16, s);
        (void)hipStreamSynchronize(s);
        if (elapsed_ms) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
        // the waiting launches go off for this model (and, process-wide, for every model created later): the next call captures the
        // five-launch step, whose kernels wait for nothing
        m->opt_fuse_attn = 0; m->opt_fuse_ffn = 0;
        drop_graphs(m);
        waits_disable("the fused QKV + attention launch timed out waiting for sibling workgroups");
        return ifa_fail(IFA_ERR_STATE, "fused decode launch: a wait for another workgroup's rows timed out (code 0x%x: 0x5_ q | k | v / attention output, 0x6_ Wo output, 0x9_ chained FFN launch); "
                        "the results of this call are not valid -- repeat it: options fuse_attn / fuse_ffn are off now (five-launch step)", (unsigned)qerr[0]);
    }
    if (elapsed_ms) { IFA_HIP_CHECK(hipEventElapsedTime(elapsed_ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
    if (out_tokens_host) memcpy(out_tokens_host, m->host_pinned + 8, sizeof(int) * (size_t)n_steps);
    th("done");
    return IFA_OK;
}

int ifa_model_decode_batch(ifa_model *m, int n, const int *tokens_host, const int *positions_host, const int *kv_slots_host,
                           int *next_tokens_host, void *logits_out_dev)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_decode_batch: model not finalized");
    IFA_REQUIRE(n >= 1 && n <= ifa_model::RING && tokens_host && positions_host && kv_slots_host, "ifa_model_decode_batch: bad arguments");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    // more queries than the fused five-launch step takes (16): balanced chunks of <= 16, each its own step (the queries are
    // independent; 32 queries op-by-op took 6.7 ms against 2 x 3.1 ms for two fused steps)
    const int fused_max = batch_fused_ok(m, 32) ? 32 : 16;
    if (n > fused_max && batch_fused_ok(m, 16)) {
        for (int c0 = 0; c0 < n; c0++) if (kv_slots_host[c0] < 0) return ifa_fail(IFA_ERR_ARG, "decode_batch: KV slot %d", kv_slots_host[c0]);
        for (int a = 0; a < n; a++)
            for (int b = 0; b < a; b++)
                if (kv_slots_host[a] == kv_slots_host[b]) return ifa_fail(IFA_ERR_ARG, "decode_batch: KV slot %d used twice", kv_slots_host[a]);
        const int k = (n + fused_max - 1) / fused_max, per = (n + k - 1) / k;
        for (int c0 = 0; c0 < n; c0 += per) {
            const int nc = std::min(per, n - c0);
            int rc = forward_batch(m, nc, tokens_host + c0, positions_host + c0, kv_slots_host + c0, next_tokens_host ? next_tokens_host + c0 : nullptr,
                                   logits_out_dev ? (char *)logits_out_dev + (size_t)c0 * m->g[T_LM_HEAD].rows * 2 : nullptr);
            if (rc) return rc;
        }
        return IFA_OK;
    }
    return forward_batch(m, n, tokens_host, positions_host, kv_slots_host, next_tokens_host, logits_out_dev);
}

int ifa_model_get_buffer(ifa_model *m, const char *name, int layer, void **dptr, size_t *bytes)
{
    IFA_REQUIRE(m && name && dptr, "ifa_model_get_buffer: null pointer");
    const ifa_model_config &c = m->cfg;
    size_t b = 0; void *p = nullptr;
    if (!strcmp(name, "logits")) { p = m->logits; b = (size_t)c.vocab * 2; }
    else if (!strcmp(name, "tp_logits")) { p = m->tp_logits; b = m->tp_logits ? m->g[T_LM_HEAD].rows * 2 : 0; }
    else if (!strcmp(name, "trace")) { p = m->trace; b = m->trace ? sizeof(long long) * 2048 * 8 : 0; }
    else if (!strcmp(name, "hidden")) { p = m->xn; b = (size_t)c.dim * 2; }
    else if (!strcmp(name, "x")) { p = m->x; b = (size_t)c.dim * 2; }
    else if (!strcmp(name, "x2")) { p = m->x2; b = (size_t)c.dim * 2; }
    else if (!strcmp(name, "dqkv")) { p = m->dqkv; b = ((size_t)c.heads + 2 * (size_t)c.kv_heads) * c.head_dim * 2; }
    else if (!strcmp(name, "att")) { p = m->att; b = (size_t)c.heads * c.head_dim * 2; }
    else if (!strcmp(name, "attq")) { p = m->attq; b = m->attq ? xq_image_bytes(c.heads * c.head_dim) : 0; }
    else if (!strcmp(name, "a")) { p = m->a; b = (size_t)c.dim * 2; }
    else if (!strcmp(name, "t1")) { p = m->t1; b = (size_t)c.ffn * 2; }
    else if (!strcmp(name, "kcache") || !strcmp(name, "vcache")) {
        IFA_REQUIRE(layer >= 0 && layer < c.layers && m->finalized, "ifa_model_get_buffer: layer %d", layer);
        p = name[0] == 'k' ? m->layers[(size_t)layer].kcache : m->layers[(size_t)layer].vcache;
        b = m->kv_row_bytes * (size_t)c.max_ctx;
    } else return ifa_fail(IFA_ERR_ARG, "ifa_model_get_buffer: unknown buffer '%s'", name);
    *dptr = p;
    if (bytes) *bytes = b;
    return IFA_OK;
}

void *ifa_model_stream(ifa_model *m) { return m ? (void *)m->stream : nullptr; }

int ifa_model_set_stream(ifa_model *m, ifa_stream stream)
{
    IFA_REQUIRE(m, "ifa_model_set_stream: null model");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    if (m->stream) IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (m->stream) (void)ifa_gemm_release_stream((ifa_stream)m->stream);
    if (m->stream && m->own_stream) IFA_HIP_CHECK(hipStreamDestroy(m->stream));
    m->stream = ifa_s(stream);
    m->own_stream = false;
    drop_graphs(m);
    return IFA_OK;
}

// ---- tensor-parallel decode, one segment per call (the caller all-reduces between them):
// the reference's DistributeAndMergeTensors sits exactly at these two seams
// (src/transformer/inference_worker.cc:1378-1391, :1882-1895).
__global__ void k_tp_set_state(int *state, int token, int pos)
{
    if (token >= 0) state[0] = token;
    if (pos >= 0) state[1] = pos;
}

// the step's token (chosen across the group) becomes the next input; the position advances on the device
// so that a captured step can be replayed (hipGraph) without the host
__global__ void k_tp_set_token(int *state, const int *token, int ring)
{
    const int t = *token;
    const int step = state[2];
    state[8 + (step % ring)] = t;       // the launch batch's token ring, like k_dec_argmax_advance
    state[0] = t;
    state[1] = state[1] + 1;
    state[2] = step + 1;
}

static int tp_ready(ifa_model *m)
{
    IFA_REQUIRE(m && m->finalized, "tensor-parallel step: model not finalized");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    std::string why;
    if (!fused_supported(m, &why)) return ifa_fail(IFA_ERR_STATE, "fused path unavailable: %s", why.c_str());
    const ifa_model_config &c = m->cfg;      // the fused epilogues apply out_scale after the LOCAL last layer: single-worker models only
    if (scale_on(c.attn_out_scale) || scale_on(c.ffn_out_scale) || scale_on(c.out_scale))
        return ifa_fail(IFA_ERR_STATE, "fused path unavailable: output scales on a partitioned model use the op-by-op path");
    return ensure_scratch(m, 1);
}

static int tp_flush_pending(ifa_model *m);

int ifa_model_tp_begin(ifa_model *m, int token, int pos)
{
    int rc = tp_ready(m);
    if (rc) return rc;
    m->pend.on = false;
    IFA_REQUIRE(pos >= -1 && pos < m->cfg.max_ctx, "ifa_model_tp_begin: position %d outside max_ctx %d", pos, m->cfg.max_ctx);
    IFA_REQUIRE(m->g[T_EMBD].present(), "ifa_model_tp_begin: this worker holds no embeddings (use ifa_model_tp_begin_hidden)");
    const ifa_model_config &c = m->cfg;
    k_tp_set_state<<<1, 1, 0, m->stream>>>(m->state, token, pos);     // token < 0: keep the id already on the device
    k_dec_gather<<<dim3(2), dim3(256), 0, m->stream>>>((const half_t *)m->g[T_EMBD].data, m->state, c.dim, (int)m->g[T_EMBD].rows,
                                                       m->x, c.rope_order ? m->rope_tab : nullptr, c.head_dim, c.rope_theta,
                                                       (int)(c.head_dim * c.partial_rotary + 0.5f), c.embd_scale);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_model_tp_begin_hidden(ifa_model *m, const void *x_f16, int pos)
{
    int rc = tp_ready(m);
    if (rc) return rc;
    IFA_REQUIRE(x_f16, "ifa_model_tp_begin_hidden: null input");
    m->pend.on = false;
    IFA_REQUIRE(pos >= -1 && pos < m->cfg.max_ctx, "ifa_model_tp_begin_hidden: position %d outside max_ctx %d", pos, m->cfg.max_ctx);
    const ifa_model_config &c = m->cfg;
    IFA_HIP_CHECK(hipMemcpyAsync(m->x, x_f16, (size_t)c.dim * 2, hipMemcpyDeviceToDevice, m->stream));
    k_tp_set_state<<<1, 1, 0, m->stream>>>(m->state, -1, pos);
    k_dec_gather<<<dim3(1), dim3(256), 0, m->stream>>>(nullptr, m->state, c.dim, 1, m->x, c.rope_order ? m->rope_tab : nullptr,
                                                       c.head_dim, c.rope_theta, (int)(c.head_dim * c.partial_rotary + 0.5f), c.embd_scale);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_model_tp_hidden(ifa_model *m, void *x_out_f16)
{
    IFA_REQUIRE(m && m->finalized && x_out_f16, "ifa_model_tp_hidden: bad arguments");
    { int rcf = tp_flush_pending(m); if (rcf) return rcf; }
    IFA_HIP_CHECK(hipMemcpyAsync(x_out_f16, m->x, (size_t)m->cfg.dim * 2, hipMemcpyDeviceToDevice, m->stream));
    return IFA_OK;
}

int ifa_model_tp_attn(ifa_model *m, int layer, void *partial_out_f16)
{
    IFA_REQUIRE(m && partial_out_f16 && layer >= 0 && layer < m->cfg.layers, "ifa_model_tp_attn: bad arguments");
    int rc;
    if ((rc = launch_qkv(m, layer, m->x))) return rc;
    if ((rc = launch_attn(m, layer))) return rc;
    return launch_wo(m, layer, m->x, (half_t *)partial_out_f16);
}

// form a pending seam sum with the op-level add kernels (when no fused consumer follows)
static int tp_flush_pending(ifa_model *m)
{
    if (!m->pend.on) return IFA_OK;
    m->pend.on = false;
    const size_t D = (size_t)m->cfg.dim;
    const half_t *src = m->pend.add;
    int rc;
    if (m->pend.bias) {
        if ((rc = ifa_add(m->pend.add, m->pend.bias, D, 0, m->f, m->stream))) return rc;
        src = m->f;
    }
    return ifa_add(m->pend.x, src, D, 0, m->pend.out, m->stream);
}

int ifa_model_tp_post_attn(ifa_model *m, int layer, const void *reduced_f16)
{
    IFA_REQUIRE(m && reduced_f16 && layer >= 0 && layer < m->cfg.layers, "ifa_model_tp_post_attn: bad arguments");
    const size_t D = (size_t)m->cfg.dim;
    const Tensor &b = m->layers[(size_t)layer].t[T_WO_B];
    const Layer &Lp = m->layers[(size_t)layer];
    if (m->cfg.parallel_attn || m->cfg.share_input) {
        // parallel attention / shared MLP input (Falcon, GPT-J/NeoX style; inference_worker.cc:847-851, 941-947): the
        // attention branch does NOT take the residual here -- attention output, FFN output and the layer input are summed
        // once after the FFN (ifa_model_tp_post_ffn).  a = merged product (+ bias once, after the merge :1388-1390)
        if (b.present()) return ifa_add(reduced_f16, b.data, D, 0, m->a, m->stream);
        IFA_HIP_CHECK(hipMemcpyAsync(m->a, reduced_f16, D * 2, hipMemcpyDeviceToDevice, m->stream));
        return IFA_OK;
    }
    // the seam sum can ride in the consumer's prologue only where that prologue exists: RMS-norm models (Std-norm ones
    // run the op-level norm kernel in front of a prologue-free GEMV)
    if (m->opt_tp_fuse_add && m->cfg.norm_kind == 0 && Lp.t[T_FFN_NORM].present() && !(m->cfg.experts > 0 && Lp.t[T_MOE_GATE].present())) {
        // a = x + (reduced + bias): left to the W1/W3 kernel's prologue (ifa_model_tp_ffn)
        m->pend.x = m->x; m->pend.add = (const half_t *)reduced_f16; m->pend.bias = (const half_t *)b.data; m->pend.out = m->a;
        m->pend.on = true;
        return IFA_OK;
    }
    const void *src = reduced_f16;
    int rc;
    if (b.present()) {     // bias once, after the merge (inference_worker.cc:1388-1390)
        if ((rc = ifa_add(reduced_f16, b.data, D, 0, m->a, m->stream))) return rc;
        src = m->a;
    }
    return ifa_add(m->x, src, D, 0, m->a, m->stream);      // Add(out, layer_input, out)
}

int ifa_model_tp_ffn(ifa_model *m, int layer, void *partial_out_f16)
{
    IFA_REQUIRE(m && partial_out_f16 && layer >= 0 && layer < m->cfg.layers, "ifa_model_tp_ffn: bad arguments");
    int rc;
    Layer &L = m->layers[(size_t)layer];
    if (m->cfg.experts > 0 && L.t[T_MOE_GATE].present()) {     // MoE: every rank routes identically (replicated gate)
        if ((rc = launch_moe_router(m, layer))) return rc;
        for (int k = 0; k < m->cfg.moe_top_k; k++) {
            if ((rc = launch_ffn13(m, layer, k))) return rc;
            if ((rc = launch_w2(m, layer, nullptr, (half_t *)partial_out_f16, k, false))) return rc;
        }
        return IFA_OK;
    }
    if ((rc = launch_ffn13(m, layer, -1, m->x))) return rc;       // (m->x: the layer input, the FFN input of shared-input models)
    return launch_w2(m, layer, nullptr, (half_t *)partial_out_f16);
}

int ifa_model_tp_post_ffn(ifa_model *m, int layer, const void *reduced_f16)
{
    IFA_REQUIRE(m && reduced_f16 && layer >= 0 && layer < m->cfg.layers, "ifa_model_tp_post_ffn: bad arguments");
    const size_t D = (size_t)m->cfg.dim;
    const Tensor &b = m->layers[(size_t)layer].t[T_W2_B];
    if (m->cfg.parallel_attn || m->cfg.share_input) {
        // next layer input = ((merged FFN product + bias) + attention output) + layer input: the order of the fused
        // single-device epilogue (residual, then residual2; inference_worker.cc:936, 941-947)
        const void *src = reduced_f16;
        int rc;
        if (b.present()) {
            if ((rc = ifa_add(reduced_f16, b.data, D, 0, m->f, m->stream))) return rc;
            src = m->f;
        }
        if ((rc = ifa_add(m->a, src, D, 0, m->f, m->stream))) return rc;
        return ifa_add(m->f, m->x, D, 0, m->x, m->stream);
    }
    if (m->opt_tp_fuse_add && m->cfg.norm_kind == 0 && layer + 1 < m->cfg.layers) {
        // next layer input = a + (reduced + bias): left to the next QKV kernel's prologue (ifa_model_tp_attn)
        m->pend.x = m->a; m->pend.add = (const half_t *)reduced_f16; m->pend.bias = (const half_t *)b.data; m->pend.out = m->x;
        m->pend.on = true;
        return IFA_OK;
    }
    const void *src = reduced_f16;
    int rc;
    if (b.present()) {
        if ((rc = ifa_add(reduced_f16, b.data, D, 0, m->f, m->stream))) return rc;
        src = m->f;
    }
    return ifa_add(src, m->a, D, 0, m->x, m->stream);       // Add(layer_out, ff_out, residual)
}

int ifa_model_tp_logits(ifa_model *m, void *logits_shard_out_f16)
{
    IFA_REQUIRE(m && logits_shard_out_f16, "ifa_model_tp_logits: bad arguments");
    IFA_REQUIRE(m->g[T_LM_HEAD].present(), "ifa_model_tp_logits: this worker holds no lm_head (not the last pipeline stage)");
    { int rcf = tp_flush_pending(m); if (rcf) return rcf; }
    return launch_lm(m, m->x, (half_t *)logits_shard_out_f16);
}

int ifa_model_tp_set_token(ifa_model *m, const int *token_dev)
{
    IFA_REQUIRE(m && token_dev, "ifa_model_tp_set_token: bad arguments");
    k_tp_set_token<<<1, 1, 0, m->stream>>>(m->state, token_dev, ifa_model::RING);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// ---- the whole multi-GPU decode step driven from C: worker segments + RCCL collectives (csrc/ifa_comm.hip) on the
// worker's stream, the distributed greedy argmax over the vocabulary-sharded lm_head, token / position fed back in
// device memory; captured once as a hipGraph and replayed per token (tensor-parallel groups; pipelines run eagerly).
static int tp_buffers(ifa_model *m)
{
    if (m->tp_a) return IFA_OK;
    const ifa_model_config &c = m->cfg;
    const size_t D = (size_t)c.dim;
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_a, D * 2));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_f, D * 2));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_hid, D * 2));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_logits, std::max<size_t>(m->g[T_LM_HEAD].rows, 1) * 2));
    return tp_argmax_scratch(m, 1);
}

// want_token = false (all but the last token of a prompt): the layers run, the lm_head / argmax / token exchange do not
static int tp_step(ifa_model *m, const ifa_tp_topology &t, int token, int pos, bool want_token = true, void *logits_copy = nullptr)
{
    const ifa_model_config &c = m->cfg;
    const size_t D = (size_t)c.dim;
    ifa_stream s = (ifa_stream)m->stream;
    const int tp_size = t.tp ? ifa_comm_size(t.tp) : 1;
    const bool merge = t.tp && (tp_size > 1 || t.force_collectives);
    int rc;
    if (t.stage == 0) { if ((rc = ifa_model_tp_begin(m, token, pos))) return rc; }
    else {
        if ((rc = ifa_recv(t.world, m->tp_hid, D * 2, t.prev_rank, s))) return rc;
        if ((rc = ifa_model_tp_begin_hidden(m, m->tp_hid, pos))) return rc;
    }
    for (int l = 0; l < c.layers; l++) {
        if ((rc = ifa_model_tp_attn(m, l, m->tp_a))) return rc;
        if (merge && (rc = ifa_allreduce_sum_f16(t.tp, m->tp_a, m->tp_a, D, s))) return rc;
        if ((rc = ifa_model_tp_post_attn(m, l, m->tp_a))) return rc;
        if ((rc = ifa_model_tp_ffn(m, l, m->tp_f))) return rc;
        if (merge && (rc = ifa_allreduce_sum_f16(t.tp, m->tp_f, m->tp_f, D, s))) return rc;
        if ((rc = ifa_model_tp_post_ffn(m, l, m->tp_f))) return rc;
    }
    if (t.n_stages > 1 && t.next_rank >= 0) {      // not the last group: hand the layer output on, then wait for the token
        if ((rc = ifa_model_tp_hidden(m, m->tp_hid))) return rc;
        if ((rc = ifa_send(t.world, m->tp_hid, D * 2, t.next_rank, s))) return rc;
        if (!want_token) return IFA_OK;
        if ((rc = ifa_broadcast(t.world, m->tp_tok, 4, t.token_src, s))) return rc;
        return ifa_model_tp_set_token(m, m->tp_tok);
    }
    if (!want_token && !logits_copy) { m->pend.on = false; return IFA_OK; }     // (the pending seam sum of the last layer is not needed)
    if ((rc = ifa_model_tp_logits(m, m->tp_logits))) return rc;
    if (logits_copy) IFA_HIP_CHECK(hipMemcpyAsync(logits_copy, m->tp_logits, m->g[T_LM_HEAD].rows * 2, hipMemcpyDeviceToDevice, m->stream));
    if (!want_token) return IFA_OK;
    if ((rc = tp_pick_rows(m, t, m->tp_logits, m->g[T_LM_HEAD].rows, (int)m->g[T_LM_HEAD].rows, 1))) return rc;
    if (t.n_stages > 1 && (rc = ifa_broadcast(t.world, m->tp_tok, 4, t.token_src, s))) return rc;
    return ifa_model_tp_set_token(m, m->tp_tok);
}

static int tp_check(ifa_model *m, const ifa_tp_topology *topo, const char *who)
{
    IFA_REQUIRE(m && topo, "%s: null pointer", who);
    const ifa_tp_topology &t = *topo;
    const int tp_size = t.tp ? ifa_comm_size(t.tp) : 1;
    IFA_REQUIRE(tp_size <= 64, "%s: group of %d ranks", who, tp_size);
    IFA_REQUIRE(t.n_stages >= 1 && t.stage >= 0 && t.stage < t.n_stages, "%s: stage %d of %d", who, t.stage, t.n_stages);
    IFA_REQUIRE(t.n_stages == 1 || t.world, "%s: layer groups need the job-wide communicator", who);
    int rc = tp_ready(m);
    if (rc) return rc;
    return tp_buffers(m);
}

// One Infer() step of a query over the partition: n_tokens new tokens at positions [start_pos, start_pos + n_tokens),
// fed through the decode path one after the other (the merges are [dim] vectors); the greedy next token of the last
// one comes back on every rank.  logits_shard_out_dev (nullable, last device group): this rank's lm_head rows of every
// token, [n_tokens][shard rows] F16 (return_output_tensors).
// A bounded wait of the one-shot exchange that gave up (a peer that never arrived) left this rank without a sum -- and its epoch
// one behind its peers'.  Every partition entry point checks after its stream sync: the call fails (the engine then aborts the
// group), the captured steps are dropped and this communicator keeps RCCL for every size from now on.
static int tp_oneshot_status(ifa_model *m, const ifa_tp_topology &t, const char *who)
{
    if (!t.tp || !ifa_comm_oneshot(t.tp)) return IFA_OK;
    const int st = ifa_comm_status(t.tp);
    if (st == 0) return IFA_OK;
    (void)ifa_comm_set_oneshot(t.tp, 0);
    drop_graphs(m);
    return ifa_fail(IFA_ERR_STATE, "%s: a wait inside the one-shot all-reduce gave up (epoch %d): a peer did not arrive; the exchange is off for this communicator", who, st);
}

int ifa_model_tp_prefill(ifa_model *m, const ifa_tp_topology *topo, const int *tokens_host, int n_tokens, int start_pos,
                         void *logits_shard_out_dev, int *next_token_host)
{
    IFA_REQUIRE(tokens_host && n_tokens >= 1, "ifa_model_tp_prefill: no tokens");
    int rc = tp_check(m, topo, "ifa_model_tp_prefill");
    if (rc) return rc;
    IFA_REQUIRE(start_pos >= 0 && start_pos + n_tokens <= m->cfg.max_ctx, "ifa_model_tp_prefill: positions %d..%d exceed max_ctx %d",
                start_pos, start_pos + n_tokens, m->cfg.max_ctx);
    if (n_tokens > 1) {
        // T > 1: the op-by-op step over all tokens at once (row-sliced GEMMs on the MFMA kernels, [T][dim] merges after
        // wo and w2, [T][dim] hand-over between layer groups) -- the reference's MatrixMultiplication branch for T > 1
        // (inference_worker.cc:2364-2432) with its merge of token_num x dim values (:2148-2195)
        m->topo = topo;
        rc = forward_ops(m, tokens_host, n_tokens, start_pos, logits_shard_out_dev, next_token_host);
        m->topo = nullptr;
        if (rc == IFA_OK) { (void)hipStreamSynchronize(m->stream); rc = tp_oneshot_status(m, *topo, "ifa_model_tp_prefill"); }
        return rc;
    }
    const size_t shard = m->g[T_LM_HEAD].present() ? m->g[T_LM_HEAD].rows * 2 : 0;
    m->host_pinned[0] = tokens_host[0]; m->host_pinned[1] = start_pos; m->host_pinned[2] = 0;
    IFA_HIP_CHECK(hipMemcpyAsync(m->state, m->host_pinned, 3 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    for (int i = 0; i < n_tokens; i++) {
        void *lg = (logits_shard_out_dev && shard) ? (char *)logits_shard_out_dev + (size_t)i * shard : nullptr;
        if ((rc = tp_step(m, *topo, tokens_host[i], start_pos + i, i + 1 == n_tokens, lg))) return rc;
    }
    IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 4, m->tp_tok, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    if ((rc = tp_oneshot_status(m, *topo, "ifa_model_tp_prefill"))) return rc;
    if (next_token_host) *next_token_host = m->host_pinned[4];
    return IFA_OK;
}

// Dynamic batching over a tensor-parallel group: one new token for each of n queries (ifa_model_decode_batch) with the
// two merges per layer over [n][dim] and one distributed argmax per row
int ifa_model_tp_decode_batch(ifa_model *m, const ifa_tp_topology *topo, int n, const int *tokens_host, const int *positions_host,
                              const int *kv_slots_host, int *next_tokens_host, void *logits_shard_out_dev)
{
    IFA_REQUIRE(n >= 1 && n <= ifa_model::RING && tokens_host && positions_host && kv_slots_host, "ifa_model_tp_decode_batch: bad arguments");
    int rc = tp_check(m, topo, "ifa_model_tp_decode_batch");
    if (rc) return rc;
    m->topo = topo;
    rc = forward_batch(m, n, tokens_host, positions_host, kv_slots_host, next_tokens_host, logits_shard_out_dev);
    m->topo = nullptr;
    if (rc == IFA_OK) { (void)hipStreamSynchronize(m->stream); rc = tp_oneshot_status(m, *topo, "ifa_model_tp_decode_batch"); }
    return rc;
}

int ifa_model_tp_decode(ifa_model *m, const ifa_tp_topology *topo, int first_token, int start_pos, int n_steps,
                        int *out_tokens_host, float *elapsed_ms)
{
    IFA_REQUIRE(out_tokens_host, "ifa_model_tp_decode: null pointer");
    IFA_REQUIRE(n_steps >= 1 && n_steps <= ifa_model::RING, "ifa_model_tp_decode: n_steps %d (1..%d)", n_steps, ifa_model::RING);
    int rc = tp_check(m, topo, "ifa_model_tp_decode");
    if (rc) return rc;
    IFA_REQUIRE(start_pos >= 0 && start_pos + n_steps <= m->cfg.max_ctx, "ifa_model_tp_decode: positions %d..%d exceed max_ctx %d",
                start_pos, start_pos + n_steps, m->cfg.max_ctx);
    const ifa_tp_topology &t = *topo;
    hipStream_t s = m->stream;
    choose_attn_split(m, start_pos + n_steps);
    // the step counter restarts: the token ring of this call begins at state[8]
    m->host_pinned[0] = first_token; m->host_pinned[1] = start_pos; m->host_pinned[2] = 0;
    IFA_HIP_CHECK(hipMemcpyAsync(m->state, m->host_pinned, 3 * sizeof(int), hipMemcpyHostToDevice, s));
    const bool use_graph = m->opt_graph && t.n_stages == 1 && (!t.tp || ifa_comm_capturable(t.tp));
    {
        ifa_model::TpKey key;
        key.tp = ifa_comm_serial(t.tp); key.world = ifa_comm_serial(t.world); key.tp_size = t.tp ? ifa_comm_size(t.tp) : 1;
        key.stage = t.stage; key.n_stages = t.n_stages; key.prev = t.prev_rank; key.next = t.next_rank; key.src = t.token_src;
        key.voff = t.vocab_offset; key.force = t.force_collectives; key.fuse = m->opt_tp_fuse_add; key.slot = m->cur_slot;
        key.oneshot = t.tp ? ifa_comm_oneshot(t.tp) : 0;
        if (m->tp_graph_exec && !(key == m->tp_key)) {
            (void)hipGraphExecDestroy(m->tp_graph_exec); m->tp_graph_exec = nullptr;
            if (m->tp_graph) { (void)hipGraphDestroy(m->tp_graph); m->tp_graph = nullptr; }
        }
        m->tp_key = key;
    }
    int done = 0;
    if (!(use_graph && m->tp_graph_exec)) {
        // the first step runs eagerly: it creates whatever the collectives allocate lazily, so that the capture below
        // records pure launches
        if ((rc = tp_step(m, t, first_token, start_pos))) return rc;
        done = 1;
        if (use_graph && done < n_steps) {      // (a one-step call has nothing to replay: no capture, no instantiate)
            IFA_HIP_CHECK(hipStreamSynchronize(s));
            IFA_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            rc = tp_step(m, t, -1, -1);
            hipGraph_t gph = nullptr;
            hipError_t e = hipStreamEndCapture(s, &gph);
            if (rc || e != hipSuccess) {
                if (gph) (void)hipGraphDestroy(gph);
                (void)hipGetLastError();
                m->tp_graph_exec = nullptr;      // eager steps below: correctness does not depend on the graph
            } else {
                m->tp_graph = gph;
                if (hipGraphInstantiate(&m->tp_graph_exec, gph, nullptr, nullptr, 0) != hipSuccess) { m->tp_graph_exec = nullptr; (void)hipGetLastError(); }
            }
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (elapsed_ms) { IFA_HIP_CHECK(hipEventCreate(&e0)); IFA_HIP_CHECK(hipEventCreate(&e1)); IFA_HIP_CHECK(hipEventRecord(e0, s)); }
    for (int i = done; i < n_steps; i++) {
        if (use_graph && m->tp_graph_exec) IFA_HIP_CHECK(hipGraphLaunch(m->tp_graph_exec, s));
        else if ((rc = tp_step(m, t, -1, -1))) return rc;
    }
    if (e1) IFA_HIP_CHECK(hipEventRecord(e1, s));
    IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 8, m->state + 8, sizeof(int) * (size_t)n_steps, hipMemcpyDeviceToHost, s));
    IFA_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < n_steps; i++) out_tokens_host[i] = m->host_pinned[8 + i];
    if (elapsed_ms) {
        float ms = 0.0f;
        IFA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *elapsed_ms = ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    // a bounded wait of the one-shot exchange that gave up (a peer that never arrived) left this rank without a sum: the
    // tokens above are not results -- fail the call (the engine then aborts the group) instead of returning them
    return tp_oneshot_status(m, t, "ifa_model_tp_decode");
}

int ifa_model_get_tensor(ifa_model *m, int layer, int tensor_id, int *dtype, void **dptr, size_t *rows, size_t *cols)
{
    IFA_REQUIRE(m && tensor_id >= 0 && tensor_id < T_MAX, "ifa_model_get_tensor: bad arguments");
    const Tensor *t;
    if (tensor_id < 10) t = &m->g[tensor_id];
    else {
        IFA_REQUIRE(layer >= 0 && layer < m->cfg.layers, "ifa_model_get_tensor: layer %d", layer);
        t = &m->layers[(size_t)layer].t[tensor_id];
    }
    if (dtype) *dtype = t->dtype;
    if (dptr) *dptr = t->data;
    if (rows) *rows = t->rows;
    if (cols) *cols = t->cols;
    return t->present() ? IFA_OK : 1;   /* 1 = tensor not set (not an error) */
}

// W1 / W2 / W3 of one expert of a mixture-of-experts layer (the layer's own tensor ids name the dense FFN)
int ifa_model_get_expert_tensor(ifa_model *m, int layer, int expert, int tensor_id, int *dtype, void **dptr, size_t *rows, size_t *cols)
{
    IFA_REQUIRE(m && layer >= 0 && layer < m->cfg.layers, "ifa_model_get_expert_tensor: layer %d", layer);
    IFA_REQUIRE(tensor_id == T_W1 || tensor_id == T_W2 || tensor_id == T_W3, "ifa_model_get_expert_tensor: tensor id %d (w1 / w2 / w3 only)", tensor_id);
    const Layer &L = m->layers[(size_t)layer];
    IFA_REQUIRE(expert >= 0 && (size_t)expert * 3 + 2 < L.experts.size(), "ifa_model_get_expert_tensor: expert %d of layer %d", expert, layer);
    const Tensor &t = L.experts[(size_t)expert * 3 + (tensor_id == T_W1 ? 0 : (tensor_id == T_W2 ? 1 : 2))];
    if (dtype) *dtype = t.dtype;
    if (dptr) *dptr = t.data;
    if (rows) *rows = t.rows;
    if (cols) *cols = t.cols;
    return t.present() ? IFA_OK : 1;
}

int ifa_model_time_kernel(ifa_model *m, int which, int iters, float *avg_us)
{
    IFA_REQUIRE(m && m->finalized && avg_us, "ifa_model_time_kernel: bad arguments");
    IFA_REQUIRE(iters > 0 && which >= 0 && which <= 9, "ifa_model_time_kernel: which %d iters %d", which, iters);
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    std::string why;
    if (!fused_supported(m, &why)) return ifa_fail(IFA_ERR_STATE, "fused path unavailable: %s", why.c_str());
    int rc = ensure_scratch(m, 1);
    if (rc) return rc;
    hipStream_t s = m->stream;
    m->host_pinned[0] = 1; m->host_pinned[1] = std::min(m->cfg.max_ctx - 1, 64); m->host_pinned[2] = 0;
    IFA_HIP_CHECK(hipMemcpyAsync(m->state, m->host_pinned, 3 * sizeof(int), hipMemcpyHostToDevice, s));
    auto touch_layer = [&](int l) {
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) {
            const Tensor &t = m->layers[(size_t)l].t[id];
            if (!t.tiled) continue;
            const size_t bytes = t.rows * ifa_tiled_row_bytes(t.dtype, t.cols);
            k_touch<<<dim3(8), dim3(256), 0, s>>>((const uint8_t *)t.tiled, bytes, (size_t)m->opt_touch_stride, m->state + 7);
        }
    };
    if (which == 7 || which == 9) {      // QKV + attention as one launch; the chained FFN launch
        if ((rc = qkv_attn_ready(m))) return rc;
        if (which == 9 && !m->ch_on) return ifa_fail(IFA_ERR_STATE, "chained FFN launch unavailable for this model / option set");
        if (which == 7 && !m->qa_on) return ifa_fail(IFA_ERR_STATE, "fused QKV + attention launch unavailable for this model / option set");
    }
    auto one = [&](int i) -> int {
        if (which == 7) return launch_qkv_attn(m, i % m->cfg.layers, m->x, (unsigned)(i + 1));
        if (which == 9) return launch_chain(m, i % m->cfg.layers, m->x, m->x2, (unsigned)(i + 1));
        const int l = m->opt_bench_mode == 1 ? 0 : i % m->cfg.layers;     // rotate over layers: distinct weights every launch
        if (m->opt_bench_mode == 2) touch_layer(l);
        switch (which) {
        case 0: return launch_qkv(m, l, m->x);
        case 1: return launch_attn(m, l);
        case 2: return launch_wo(m, l, m->x);
        case 3: return launch_ffn13(m, l, -1, m->x);
        case 4: return launch_w2(m, l, m->x2);
        default: return launch_lm(m, m->x);
        }
    };
    k_dec_gather<<<dim3(2), dim3(256), 0, s>>>((const half_t *)m->g[T_EMBD].data, m->state, m->cfg.dim, (int)m->g[T_EMBD].rows, m->x,
                                               m->cfg.rope_order ? m->rope_tab : nullptr, m->cfg.head_dim, m->cfg.rope_theta,
                                               (int)(m->cfg.head_dim * m->cfg.partial_rotary + 0.5f), m->cfg.embd_scale);
    for (int i = 0; i < 3; i++) if ((rc = one(i))) return rc;
    if (m->opt_trace) {
        if (!m->trace) IFA_HIP_CHECK(hipMalloc((void **)&m->trace, sizeof(long long) * 2048 * 8));
        IFA_HIP_CHECK(hipMemsetAsync(m->trace, 0, sizeof(long long) * 2048 * 8, s));
        g_trace_ptr = m->trace;
    }
    hipEvent_t e0, e1;
    IFA_HIP_CHECK(hipEventCreate(&e0)); IFA_HIP_CHECK(hipEventCreate(&e1));
    IFA_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++) if ((rc = one(i))) return rc;
    IFA_HIP_CHECK(hipEventRecord(e1, s));
    IFA_HIP_CHECK(hipStreamSynchronize(s));
    g_trace_ptr = nullptr;
    float ms = 0;
    IFA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = ms * 1000.0f / (float)iters;
    return IFA_OK;
}

} // extern "C"
