// ifa_engine_state.h -- the per-device worker's state (struct ifa_model: tensors, scratch, captured steps, options) and the
// functions its translation units share.  The worker is the MI355X counterpart of GpuInferenceWorker
// (src/transformer/inference_worker.cc:234-340); it is split by what a unit does, not by layer:
//   ifa_engine.hip          model life cycle: create / set_tensor / finalize / KV slots / options / buffers (the C ABI's ifa_model_* basics)
//   ifa_engine_decode.hip   the fused batch-1 decode step: launch parameters of every fused kernel, graph capture, ifa_model_decode
//   ifa_engine_forward.hip  prompts and batched steps: op-by-op layer, the four-launch prompt routes, batched steps, ifa_model_forward
//   ifa_engine_moe.hip      mixture of experts: the router on the device, grouped expert launches of a batch, the host-routed fallback
//   ifa_engine_exact.hip    option exact_order: single-token steps in the reference kernels' summation order (parity instrument)
//   ifa_engine_tp.hip       tensor / layer partitions: per-seam entry points and the multi-GPU step driven from C
#pragma once
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

namespace ifae {


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
    void *q3hn = nullptr;    // Q3H_B64T1: the values at the format's native 32 bytes per block for the decode GEMVs (option q3h_native, built on first use)
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


} // namespace ifae
using namespace ifae;

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
    int opt_attn_unload = 1;       // 1 (default): in the 256-row bucket the heads' workgroups of the fused QKV + attention launch take no weight rows (UL kernels)
    int opt_attn_nsplits = 0;      // > 0: splits per head whatever the context (measurement)
    // attention as the tail of the QKV launch (ifa_decode_qkv_attn.h): granules [layers][(heads + 2 kv_heads) * head_dim], the
    // decode-call counter the tags are built from, its own error word
    int opt_fuse_attn = 1, opt_fuse_attn_timeout_us = 20000, qa_on = 0, qa_gk = 0;
    unsigned long long *qa_gran = nullptr;
    unsigned *qa_call = nullptr, *qa_err = nullptr, qa_calls = 0;
    // consecutive GEMV ops of a layer as ONE launch with the next op's rows requested before the hand-off (ifa_decode_chain.h):
    // option fuse_ffn = 1: W1 | W3 -> W2; 2: Wo -> W1 | W3 -> W2.  ch_on = what the captured step uses.  Granules [layers][dim + ffn].
    int opt_fuse_ffn = 0, ch_on = 0, opt_chain_late_w2 = 0;
    int opt_q3h_native = 0;          // Q3H_B64T1 linears of the dense FFN and Wo streamed at 32 instead of 36 bytes per block (A / B: profiles/r06_q3h_native_ab.log)
    // Per-phase times in the reference's key space (InferencePerfStat, GpuInferenceWorker::UpdatePerfStat, inference_worker.cc:2670-2697:
    // (layer + 1) * 10000 + phase, the whole layer under + 0 for layers 0..5, the phases of layer 0 -- layer_idx_for_study_ -- under
    // + 10 / 30 / 50 / 60 / 90 / 300 / 700 / 710 / 730 / 750 / 760 / 780 / 800, the output stage under 1000009, the embedding rows under 1).
    // Option perf_stat = 1: steps and prompts take the op-by-op layer (one launch per reference op) with a HIP event pair around every
    // phase; ifa_model_perf_stat reads the accumulated milliseconds.  The reference times the host side of the launches (TaskMonitor);
    // these are device times of the same spans.
    int opt_perf_stat = 0;
    struct PerfSpanRec { int key; hipEvent_t e0, e1; };
    std::vector<PerfSpanRec> perf_spans;
    std::vector<hipEvent_t> perf_pool;
    std::map<int, float> perf_map;
    int opt_exact_order = 0;         // single-token steps in the reference kernels' summation order (ifa_engine_exact.hip): bit-identical to the oracle
    float *exact_rope_tab = nullptr; // device: [max_ctx][head_dim / 2] (cos, sin) from the host libm, built on the first exact step
    uint32_t *ch_gran = nullptr, *ch_flags = nullptr;      // flags [layers][2][CH_FLAGS]
    // the end of the step as one launch (ifa_decode_lmhead_tail.h): lm_head + argmax + state advance + the next step's gather.
    // st_on = what the captured step uses (F16 lm_head with the RMS / no final norm)
    int opt_step_tail = 1, st_on = 0;
    unsigned long long *st_keys = nullptr; unsigned *st_counter = nullptr; int st_keys_n = 0;
    int attn_pb = 256, opt_attn_kt = 1;      // cache rows the one-workgroup decode attention requests at entry (64 / 128 / 256: the bucket the call stays inside); K rows through the LDS tile
    int attn_split = 0, opt_attn_split_ctx = -1, opt_batch_graph = 0, opt_gemm_rows = 1, opt_batch_fused = 1, opt_moe_router_fused = 1, opt_prefill_big = 1, opt_rows_mo = 1, opt_moe_singles = 1;
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
    // round 6: prompts of 33 .. prefill_mid_max tokens take the four launches per layer from k_gemm_mid (ifa_gemm_mid.hip: ring of
    // direct-to-LDS stages requested by loader waves, weights dequantised into the MFMA operand registers) when every linear has its
    // operand-order copy; 320..768 tokens 8-9 % faster than the large tiles, 1024 tokens a tie (profiles/r06_prefill_mid_parts.log)
    int opt_prefill_mid = 1, opt_prefill_mid_max = 768;
    int opt_prefill_res_mid = 2048;   // prefill_mid_max + 1 .. this many tokens: wo / w2 still through k_gemm_mid, the other products through the large tiles (0: never)
    int opt_rows_kparts = 1, opt_gemm_splitk = 1;   // 0: never the launches whose workgroups wait for partner workgroups (K parts of the 9..32-row GEMM, split-K halves of the large-tile GEMM)
    int opt_debug_mo_alloc_fail = 0;           // tests: ensure_mo_build fails like an exhausted allocator after its first copy
    static constexpr int RING = 1024;
};


static inline bool is_q4(int dt) { return dt == Q4_B32T1A || dt == Q4_B32T1B; }

// formats the MO copy of the rows GEMM takes: 4-bit codes with value q * scale + base (the 64-weight ones only through MO)
static inline bool rows_mo_fmt(int dt) { return is_q4(dt) || dt == Q4_B64T1 || dt == Q3H_B64T1; }

static inline bool scale_on(float s) { return s < 0.9999f || s > 1.0001f; }     // the reference's test for "scale != 1"

// same tiled layout and arithmetic (the A/B variants differ only in how the quantizer picked base/scale)
static inline bool same_fmt(int a, int b) { return a == b || (is_q4(a) && is_q4(b)); }


static constexpr size_t IFA_LDS_LIMIT = 160 * 1024;      // LDS per workgroup on gfx950 (MI355X_MICROARCH.md)

static inline int num_cus() { return dec_num_cus(); }


// The fused GEMV of a weight tensor: int8-path formats stream their tiled copy (k_dec_gemv), everything else -- F16
// tensors, Q8_B32T1 / Q5_B32T1 / Q4_B16 / Q3_B32T1 / Q2_B32T1 -- the reference-layout bytes with fp16 activations
// (k_dec_gemv_h).  wbytes() hands out the matching pointer.
static inline bool fused_int8(int w_dtype) { return ax8_eligible(w_dtype); }

static inline const uint8_t *wbytes(const Tensor &t) { return (const uint8_t *)(fused_int8(t.dtype) ? t.tiled : t.data); }


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

// weight pointer of a rows-GEMM launch: the MO copy when it exists (all sets of a launch alike: ensure_mo builds all or none)
static inline const uint8_t *rows_w(const ifa_model *m, const Tensor &t) { return (const uint8_t *)(m->opt_rows_mo && t.mo ? t.mo : t.tiled); }

static inline int rows_mo(const ifa_model *m, const Tensor &t) { return m->opt_rows_mo && t.mo ? 1 : 0; }

namespace ifae {

// ---- ifa_engine.hip
void drop_graphs(ifa_model *m);
// perf_stat spans (ifa_engine.hip): no-ops unless the option is on.  PerfSpan brackets a phase on the model's stream.
int perf_begin(ifa_model *m, int key);
void perf_end(ifa_model *m, int idx);
int perf_collect(ifa_model *m);
struct PerfSpan {
    ifa_model *m; int idx;
    PerfSpan(ifa_model *mm, int key, bool on = true) : m(mm), idx(on ? perf_begin(mm, key) : -1) {}
    ~PerfSpan() { done(); }
    void done() { if (idx >= 0) { perf_end(m, idx); idx = -1; } }
};
void free_tensor(Tensor &t);
void *kv_ptr(ifa_model *m, size_t layer, int slot, bool is_v);
int ensure_mo(ifa_model *m);
int ensure_mo_build(ifa_model *m);
int ensure_x32(ifa_model *m);
int ensure_q3hn(ifa_model *m);
int ensure_scratch(ifa_model *m, int T);
// ---- ifa_engine_decode.hip
void choose_attn_split(ifa_model *m, int reach);
bool fused_ok(const Tensor &t, bool long_rows);
int lmhead_grid(const DecLmHeadParams &P, int wgs_per_cu_opt);
int launch_lmhead(const DecLmHeadParams &P, int norm, int wgs_per_cu_opt, hipStream_t s, const DecStepTail *Z = nullptr);
bool has_post_norms(const ifa_model *m);
bool fused_supported(const ifa_model *m, std::string *why);
int sep_norm(ifa_model *m, const half_t *x, const Tensor &w, const Tensor &b, half_t *dst);
void attn_params(ifa_model *m, int l, DecAttnParams &A);
bool qkv_attn_layer_ok(const ifa_model *m, int l, int *gk_out);
int qkv_attn_ready(ifa_model *m);
void qkv_params(ifa_model *m, int l, const half_t *x, DecGemvParams &P);
int launch_qkv_attn(ifa_model *m, int l, const half_t *x, unsigned tag_add = 0);
int launch_qkv(ifa_model *m, int l, const half_t *x);
int launch_attn(ifa_model *m, int l);
int launch_wo(ifa_model *m, int l, const half_t *x, half_t *partial = nullptr);
void moe_params(ifa_model *m, Layer &L, DecGemvParams &P, int slot, int tab_off);
int launch_ffn13(ifa_model *m, int l, int moe_slot = -1, const half_t *x_layer = nullptr, int moe_nslots = 1);
int launch_w2(ifa_model *m, int l, half_t *xnext, half_t *partial = nullptr, int moe_slot = -1, bool moe_last = false,
                     const half_t *residual2 = nullptr, int moe_t1_slot = 0);
int launch_chain(ifa_model *m, int l, const half_t *x, half_t *xnext, unsigned tag_add = 0);
bool step_tail_ok(const ifa_model *m);
DecLmHeadParams lm_params(ifa_model *m, const half_t *x, half_t *logits_out);
int step_tail_ready(ifa_model *m);
int launch_lm_tail(ifa_model *m, const half_t *x);
int launch_gather(ifa_model *m);
int launch_lm(ifa_model *m, const half_t *x, half_t *logits_out = nullptr);
int enqueue_fused_step(ifa_model *m);
int decode_impl(ifa_model *m, int first_token, int start_pos, int n_steps, int *out_tokens_host, float *elapsed_ms, bool prepare_only);
// ---- ifa_engine_forward.hip
int gather_rows(ifa_model *m, const half_t *src, const int *idx_dev, int T, int dim, int n_src, half_t *dst, float scale);
int matmul(ifa_model *m, const half_t *A, int T, const Tensor &W, const Tensor &bias, half_t *C);
int norm_rows(ifa_model *m, const half_t *x, int T, const Tensor &w, const Tensor &b, half_t *y, float base = 0.0f);
int ffn_dense(ifa_model *m, const half_t *x, int T, const Tensor &w1, const Tensor &b1, const Tensor &w3, const Tensor &b3,
                     const Tensor &w2, const Tensor &b2, half_t *out, int perf_base = 0);
int layer_tail_ops(ifa_model *m, int l, int T, half_t *&x, const half_t *attn_in, bool &xn_ready);
int forward_ops(ifa_model *m, const int *tokens_host, int T, int prefix_len, void *logits_out, int *next_token, bool no_head = false);
bool batch_fused_ok(const ifa_model *m, int n);
bool prefill_big_ok(const ifa_model *m);
bool prefill_mid_ok(ifa_model *m, int T, bool any_length);
int batch_fused_layer(ifa_model *m, int l, int n, const half_t *x, half_t *xnext, const void *rows_l);
int forward_batch(ifa_model *m, int n, const int *tokens_host, const int *pos_host, const int *slot_host, int *next_tokens,
                         void *logits_out);
// ---- ifa_engine_moe.hip
int launch_moe_router(ifa_model *m, int l);
// Mixture of experts (ProcessGpuLayer_Moe, inference_worker.cc:1924-2146): router GEMV -> softmax -> D2H ->
// host top-k (HostTensorOpr::BuildRowsForMoE, host_tensor_opr.cc:190-244: probabilities below 1e-5 are dropped,
// optional renormalisation) -> the selected experts' FFNs in ascending expert order, each row on the T=1
// path -> B[row] = hfma(out, weight, B[row]) (AddByRowIdx_Kernel).  Result in m->f.
bool moe_device_ok(const ifa_model *m, const Layer &L);
int moe_ffn_device(ifa_model *m, Layer &L, const half_t *ff_n, int T, const half_t *pre_norm = nullptr, const half_t *residual = nullptr, half_t *out = nullptr);
int moe_ffn(ifa_model *m, Layer &L, const half_t *ff_n, int T);
int max_smalls_possible(bool rows_kernel, int E, int cap);
int ensure_side_stream(ifa_model *m);
bool moe_router_rows_ok(const ifa_model *m, const Layer &L, int T);
// ---- ifa_engine_exact.hip
bool exact_supported(const ifa_model *m, std::string *why);
int forward_exact(ifa_model *m, int token, int pos, void *logits_out, int *next_token);
// ---- ifa_engine_tp.hip
int tp_argmax_scratch(ifa_model *m, size_t n_rows);
int tp_pick_rows(ifa_model *m, const ifa_tp_topology &t, const half_t *shard, size_t row_stride, int shard_rows, int n_rows);
bool tp_merging(const ifa_model *m);
int tp_merge_rows(ifa_model *m, half_t *buf, int T, const Tensor &bias);
int tp_ready(ifa_model *m);
int tp_flush_pending(ifa_model *m);
int tp_buffers(ifa_model *m);
int tp_step(ifa_model *m, const ifa_tp_topology &t, int token, int pos, bool want_token = true, void *logits_copy = nullptr);
int tp_check(ifa_model *m, const ifa_tp_topology *topo, const char *who);
int tp_oneshot_status(ifa_model *m, const ifa_tp_topology &t, const char *who);

} // namespace ifae
