// ifa_engine.hip -- per-device decode worker, model life cycle (see ifa_engine_state.h for the layout of the worker's units):
// tensors in (reference layout + the re-tiled / MFMA-operand-order copies), KV caches and query slots, options, debug buffers.
#include "ifa_engine_state.h"

namespace ifae {

int perf_begin(ifa_model *m, int key)
{
    if (!m->opt_perf_stat) return -1;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(m->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return -1; }
    auto take = [&]() -> hipEvent_t {
        if (!m->perf_pool.empty()) { hipEvent_t e = m->perf_pool.back(); m->perf_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return e;
    };
    ifa_model::PerfSpanRec r; r.key = key; r.e0 = take(); r.e1 = take();
    if (!r.e0 || !r.e1) { if (r.e0) m->perf_pool.push_back(r.e0); if (r.e1) m->perf_pool.push_back(r.e1); return -1; }
    if (hipEventRecord(r.e0, m->stream) != hipSuccess) { (void)hipGetLastError(); m->perf_pool.push_back(r.e0); m->perf_pool.push_back(r.e1); return -1; }
    m->perf_spans.push_back(r);
    return (int)m->perf_spans.size() - 1;
}

void perf_end(ifa_model *m, int idx)
{
    if (idx < 0 || idx >= (int)m->perf_spans.size()) return;
    if (hipEventRecord(m->perf_spans[(size_t)idx].e1, m->stream) != hipSuccess) (void)hipGetLastError();
}

// the stream is drained, every span's milliseconds are ADDED to its key (UpdatePerfStat's rule: iter->second += value)
int perf_collect(ifa_model *m)
{
    if (m->perf_spans.empty()) return IFA_OK;
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    for (auto &r : m->perf_spans) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) m->perf_map[r.key] += ms;
        else (void)hipGetLastError();
        m->perf_pool.push_back(r.e0); m->perf_pool.push_back(r.e1);
    }
    m->perf_spans.clear();
    return IFA_OK;
}

void drop_graphs(ifa_model *m)
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

void free_tensor(Tensor &t)
{
    if (t.data) (void)hipFree(t.data);
    if (t.tiled) (void)hipFree(t.tiled);
    if (t.mo) (void)hipFree(t.mo);
    if (t.x32) (void)hipFree(t.x32);
    if (t.q3hn) (void)hipFree(t.q3hn);
    t = Tensor();
}

// ------------------------------------------------ op-by-op forward (any T)
int ensure_scratch(ifa_model *m, int T)
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

void *kv_ptr(ifa_model *m, size_t layer, int slot, bool is_v)
{
    if (slot == m->cur_slot || m->slots.empty()) return is_v ? m->layers[layer].vcache : m->layers[layer].kcache;
    const ifa_model::KvSlot &sl = m->slots[(size_t)slot];
    return is_v ? sl.v[layer] : sl.k[layer];
}

// (a failed allocation -- the copies are 4.3 GB for Llama-2-7B Q4 -- is not an error of the step that triggered it: every partial
//  copy is dropped, opt_rows_mo goes off and the tiled kernels (<= 16 rows per launch) serve the batched steps)
int ensure_mo(ifa_model *m)
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

int ensure_mo_build(ifa_model *m)
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
int ensure_x32(ifa_model *m)
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

// option q3h_native: the 32-byte-per-block streaming copies of the Q3H_B64T1 matrices the k_dec_gemv launches of a dense layer read
// (Wo, W1, W3, W2), built on the first decode call with the option on
int ensure_q3hn(ifa_model *m)
{
    bool built = false;
    for (Layer &L : m->layers) {
        const int ids[] = {T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) {
            Tensor &t = L.t[id];
            if (t.q3hn || !t.present() || t.dtype != Q3H_B64T1 || t.cols % 64 != 0) continue;
            IFA_HIP_CHECK(hipMalloc(&t.q3hn, t.rows * (t.cols / 64) * 32));
            int rc = q3h_native_rows(t.data, t.rows, t.cols, t.q3hn, m->stream);
            if (rc) return rc;
            built = true;
        }
    }
    if (built) IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    return IFA_OK;
}

} // namespace ifae

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
    if (m->exact_rope_tab) (void)hipFree(m->exact_rope_tab);
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
    for (auto &r : m->perf_spans) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (hipEvent_t e : m->perf_pool) (void)hipEventDestroy(e);
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
    IFA_REQUIRE(block_capacity(dtype) > 0 && dtype != F32 && dtype != Q3H_NATIVE, "ifa_model_set_tensor: dtype %d", dtype);
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
        IFA_HIP_CHECK(hipMalloc((void **)&m->attn_ws.S, (size_t)c.heads * c.max_ctx * 2 + 64));      // (+ 64: the last row is read in whole 16-byte pieces)
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
        {"bench_mode", &m->opt_bench_mode}, {"touch_stride", &m->opt_touch_stride}, {"attn_split_ctx", &m->opt_attn_split_ctx}, {"batch_graph", &m->opt_batch_graph}, {"gemm_rows", &m->opt_gemm_rows}, {"batch_fused", &m->opt_batch_fused}, {"rows_mo", &m->opt_rows_mo}, {"debug_mo_alloc_fail", &m->opt_debug_mo_alloc_fail}, {"rows_kparts", &m->opt_rows_kparts}, {"prefill_chunk", &m->opt_prefill_chunk}, {"prefill_big_min", &m->opt_prefill_big_min}, {"prefill_mid", &m->opt_prefill_mid}, {"prefill_mid_max", &m->opt_prefill_mid_max}, {"prefill_res_mid", &m->opt_prefill_res_mid}, {"gemm_splitk", &m->opt_gemm_splitk}, {"moe_singles", &m->opt_moe_singles}, {"moe_overlap", &m->opt_moe_overlap}, {"prefill_big", &m->opt_prefill_big}, {"moe_router_fused", &m->opt_moe_router_fused}, {"tp_fuse_add", &m->opt_tp_fuse_add}, {"attn_q8", &m->opt_attn_q8}, {"attn_kt", &m->opt_attn_kt}, {"fuse_attn", &m->opt_fuse_attn}, {"fuse_ffn", &m->opt_fuse_ffn}, {"attn_post_as_residual", &m->opt_attn_post_as_residual}, {"chain_late_w2", &m->opt_chain_late_w2}, {"exact_order", &m->opt_exact_order}, {"perf_stat", &m->opt_perf_stat}, {"attn_nsplits", &m->opt_attn_nsplits}, {"attn_unload", &m->opt_attn_unload}, {"q3h_native", &m->opt_q3h_native}, {"fuse_attn_timeout_us", &m->opt_fuse_attn_timeout_us}, {"step_tail", &m->opt_step_tail}, {"graph_steps", &m->opt_graph_steps}, {"moe_device", &m->opt_moe_device}, 
        
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
// (layer + 1) * 10000 + phase -> milliseconds accumulated since the last clear (ascending keys); see opt_perf_stat
int ifa_model_perf_stat(ifa_model *m, int *keys_out, float *ms_out, int cap, int *n_out, int clear)
{
    IFA_REQUIRE(m && n_out && cap >= 0 && (cap == 0 || (keys_out && ms_out)), "ifa_model_perf_stat: null pointer");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    int rc = perf_collect(m);
    if (rc) return rc;
    int n = 0;
    for (auto &kv : m->perf_map) {
        if (n < cap) { keys_out[n] = kv.first; ms_out[n] = kv.second; }
        n++;
    }
    *n_out = n;                      // (the number of keys there are: a caller with a small buffer asks again)
    if (clear) m->perf_map.clear();
    return IFA_OK;
}

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

} // extern "C"
