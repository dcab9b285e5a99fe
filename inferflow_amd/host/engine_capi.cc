// engine_capi.cc -- C ABI of the InferenceEngine facade (include/inferflow_engine.h)
#include <algorithm>
#include <chrono>
#include <cstring>
#include <future>
#include <map>
#include <string>

#include "inferflow_engine.h"
#include "inference_engine.h"
#include "inferflow_service.h"
#include "perplexity.h"

using namespace inferflow_amd;

struct ifa_engine {
    InferenceEngine engine;
    std::map<int, QueryInferenceResult> last;     // per query: result of the most recent step
    InferencePerfStat last_perf;                  // perf_stat of the most recent Infer()
};

extern "C" {

const char *ifa_engine_last_error(void) { return EngineLastError(); }

ifa_engine *ifa_engine_create(const char *ini_path, const char *section, const char *data_root_dir)
{
    if (!ini_path || !section) { EngineSetError("ifa_engine_create: null argument"); return nullptr; }
    InferenceConfig cfg;
    if (!InferenceEngine::LoadConfig(cfg, ini_path, section, data_root_dir ? data_root_dir : "")) return nullptr;
    ifa_engine *e = new ifa_engine();
    if (!e->engine.Init(cfg)) { delete e; return nullptr; }
    return e;
}

void ifa_engine_destroy(ifa_engine *e) { delete e; }

int ifa_engine_add_query(ifa_engine *e, const int *tokens, int n_tokens)
{
    if (!e || !tokens || n_tokens <= 0) { EngineSetError("ifa_engine_add_query: bad arguments"); return -1; }
    return e->engine.AddQuery(std::vector<int>(tokens, tokens + n_tokens), QueryOptions());
}

int ifa_engine_add_query_ex(ifa_engine *e, const int *tokens, int n_tokens, int strategy_id, int random_seed, float temperature)
{
    if (!e || !tokens || n_tokens <= 0) { EngineSetError("ifa_engine_add_query_ex: bad arguments"); return -1; }
    QueryOptions opt; opt.strategy_id = strategy_id; opt.random_seed = random_seed; opt.temperature = temperature;
    return e->engine.AddQuery(std::vector<int>(tokens, tokens + n_tokens), opt);
}

int ifa_engine_strategy_id(ifa_engine *e, const char *name)
{
    return (int)(e ? e->engine.GetSamplingStrategyId(name ? name : "") : SamplingStrategyIdFromName(name ? name : ""));
}

int ifa_sampling_choose(const uint16_t *logits_f16, int vocab, int strategy_id, int max_k, float top_p, int pool_size,
                        float temperature, long long seed, int n_draws, int *out_ids, float *out_probs,
                        int *pool_ids, float *pool_probs, int pool_capacity)
{
    if (!logits_f16 || vocab <= 0 || n_draws < 0) { EngineSetError("ifa_sampling_choose: bad arguments"); return -1; }
    StdSamplingConfig cfg; cfg.max_k = max_k; cfg.top_p = top_p; cfg.pool_size = pool_size;
    JavaRandom rng((uint64_t)seed);
    int pool_n = 0;
    for (int d = 0; d < std::max(1, n_draws); d++) {
        SamplingOutput out; SamplingState st;
        if (!IsStdFamily((SamplingStrategyId)strategy_id) || !ChooseTokens(out, logits_f16, vocab, (SamplingStrategyId)strategy_id, cfg, temperature, rng, st)) { EngineSetError("ifa_sampling_choose: unsupported strategy %d", strategy_id); return -1; }
        if (d < n_draws && !out.selected.empty()) { if (out_ids) out_ids[d] = out.selected[0].id; if (out_probs) out_probs[d] = out.selected[0].weight; }
        pool_n = (int)out.token_pool.size();
        for (int i = 0; i < pool_n && i < pool_capacity; i++) { if (pool_ids) pool_ids[i] = out.token_pool[(size_t)i].id; if (pool_probs) pool_probs[i] = out.token_pool[(size_t)i].weight; }
    }
    return pool_n;
}

double ifa_perplexity_token_nll(const uint16_t *logits_f16, int vocab, int token_id)
{
    if (!logits_f16 || vocab <= 0 || token_id < 0 || token_id >= vocab) { EngineSetError("ifa_perplexity_token_nll: bad arguments"); return -1.0; }
    return TokenNll(logits_f16, vocab, token_id);
}

int ifa_sampling_choose_ex(const uint16_t *logits_f16, int vocab, int strategy_id, const float *params9, float temperature,
                           long long seed, int n_draws, int *out_ids, float *out_probs, int *pool_ids, float *pool_probs,
                           int pool_capacity, float *mirostat_mu_inout, const int *text_tokens, int n_text)
{
    if (!logits_f16 || vocab <= 0 || n_draws < 0 || !params9) { EngineSetError("ifa_sampling_choose_ex: bad arguments"); return -1; }
    SamplingConfig cfg;
    cfg.max_k = (int)params9[0]; cfg.top_p = params9[1]; cfg.pool_size = (int)params9[2]; cfg.min_p = params9[3]; cfg.tfs_z = params9[4];
    cfg.typical_p = params9[5]; cfg.mirostat_eta = params9[6]; cfg.mirostat_tau = params9[7]; cfg.eos_bypassing_max = (int)params9[8];
    cfg.rfsd_top_p = cfg.top_p;            // (RandomizedFSD's sampling branch has its own top_p: 0.93 by default, here the caller's)
    JavaRandom rng((uint64_t)seed);
    SamplingState st;
    if (mirostat_mu_inout) st.mirostat_mu = *mirostat_mu_inout;
    const std::vector<int> text(text_tokens, text_tokens + (text_tokens ? std::max(0, n_text) : 0));
    int pool_n = 0;
    for (int d = 0; d < std::max(1, n_draws); d++) {
        SamplingOutput out;
        if (!ChooseTokens(out, logits_f16, vocab, (SamplingStrategyId)strategy_id, cfg, temperature, rng, st, text)) { EngineSetError("ifa_sampling_choose_ex: unsupported strategy %d", strategy_id); return -1; }
        if (d < n_draws && !out.selected.empty()) { if (out_ids) out_ids[d] = out.selected[0].id; if (out_probs) out_probs[d] = out.selected[0].weight; }
        pool_n = (int)out.token_pool.size();
        for (int i = 0; i < pool_n && i < pool_capacity; i++) { if (pool_ids) pool_ids[i] = out.token_pool[(size_t)i].id; if (pool_probs) pool_probs[i] = out.token_pool[(size_t)i].weight; }
    }
    if (mirostat_mu_inout) *mirostat_mu_inout = st.mirostat_mu;
    return pool_n;
}

int ifa_sampling_random_doubles(long long seed, int n, double *out)
{
    if (!out || n < 0) return 0;
    JavaRandom rng((uint64_t)seed);
    for (int i = 0; i < n; i++) out[i] = rng.NextDouble();
    return n;
}

int ifa_engine_query_count(ifa_engine *e) { return e ? e->engine.QueryCount() : -1; }

int ifa_engine_remove_query(ifa_engine *e, int query_id)
{
    if (!e) return 0;
    e->last.erase(query_id);
    return e->engine.RemoveQuery(query_id) ? 1 : 0;
}

int ifa_engine_infer(ifa_engine *e, int *query_ids, int *next_tokens, int capacity)
{
    if (!e) { EngineSetError("ifa_engine_infer: null engine"); return -1; }
    InferenceResult res;
    if (!e->engine.Infer(res)) return -1;
    e->last_perf = res.perf_stat;
    int n = 0;
    for (QueryInferenceResult &item : res.items) {
        if (n < capacity && query_ids && next_tokens) { query_ids[n] = item.query_id; next_tokens[n] = item.next_tokens.empty() ? -1 : item.next_tokens[0].id; }
        n++;
        e->last[item.query_id] = std::move(item);
    }
    return n;
}

int ifa_engine_perf_stat(ifa_engine *e, unsigned *keys, float *ms, int capacity)
{
    if (!e) { EngineSetError("ifa_engine_perf_stat: null engine"); return -1; }
    int n = 0;
    for (const auto &kv : e->last_perf.time_map) {
        if (n < capacity && keys && ms) { keys[n] = kv.first; ms[n] = kv.second; }
        n++;
    }
    return n;
}

int ifa_engine_commit(ifa_engine *e, const int *query_ids, const int *tokens, const int *is_end, int n)
{
    if (!e || !query_ids || !tokens || n < 0) { EngineSetError("ifa_engine_commit: bad arguments"); return 0; }
    std::map<int, QueryNextToken> m;
    for (int i = 0; i < n; i++) { QueryNextToken t; t.id = tokens[i]; t.is_end = is_end && is_end[i]; m[query_ids[i]] = t; }
    return e->engine.CommitInferenceResult(m) ? 1 : 0;
}

int ifa_engine_last_logits(ifa_engine *e, int query_id, uint16_t *dst_f16, size_t capacity, int *rows, int *cols)
{
    if (!e) return 0;
    auto it = e->last.find(query_id);
    if (it == e->last.end()) { EngineSetError("no result for query %d", query_id); return 0; }
    if (rows) *rows = it->second.output_rows;
    if (cols) *cols = it->second.output_cols;
    if (dst_f16) memcpy(dst_f16, it->second.output_tensor.data(), std::min(capacity, it->second.output_tensor.size()) * 2);
    return 1;
}

int ifa_engine_generate(ifa_engine *e, int query_id, int n_steps, int *out_tokens, float *gpu_ms)
{
    if (!e || !out_tokens) { EngineSetError("ifa_engine_generate: bad arguments"); return -1; }
    std::vector<int> toks;
    if (!e->engine.Generate(query_id, n_steps, toks, gpu_ms)) return -1;
    for (size_t i = 0; i < toks.size(); i++) out_tokens[i] = toks[i];
    return (int)toks.size();
}

int ifa_engine_perplexity(ifa_engine *e, const int *tokens, int n_tokens, int max_length, int stride,
                          double *ppl, double *ppl_stderr, long long *count)
{
    if (!e || !tokens || n_tokens <= 0) { EngineSetError("ifa_engine_perplexity: bad arguments"); return 0; }
    PerplexityResult r;
    if (!ComputePerplexity(e->engine, std::vector<int>(tokens, tokens + n_tokens), max_length, stride, r)) return 0;
    if (ppl) *ppl = r.ppl;
    if (ppl_stderr) *ppl_stderr = r.ppl_stderr;
    if (count) *count = r.count;
    return 1;
}

int ifa_engine_model_info(ifa_engine *e, const char *key)
{
    if (!e || !key) return -1;
    const ModelSpec &s = e->engine.model_spec();
    const std::string k = key;
    if (k == "vocab_size") return s.hyper_params.vocab_size;
    if (k == "embd_dims") return s.hyper_params.embd_dims;
    if (k == "hidden_dim") return s.hyper_params.hidden_dim;
    if (k == "decoder_layers") return s.hyper_params.decoder_layers;
    if (k == "decoder_heads") return s.hyper_params.decoder_heads;
    if (k == "decoder_kv_heads") return s.hyper_params.decoder_kv_heads;
    if (k == "max_context_len") return s.max_context_len > 0 ? s.max_context_len : ModelSpec::DEFAULT_MAX_CONTEXT_LEN;
    if (k == "device_weight_data_type") return s.device_weight_data_type;
    if (k == "device_kv_cache_data_type") return s.device_kv_cache_data_type;
    if (k == "partition_ranks") return e->engine.PartitionRanks();
    return -1;
}

// the worker (ifa_model *, for the ifa_model_* calls of include/inferflow_amd.h) of partition rank `rank` and its plan
// {stage, n_stages, tp_rank, tp_size, layer0, layer1}; null / -1 for a rank that does not exist
void *ifa_engine_worker(ifa_engine *e, int rank) { return e ? (void *)e->engine.worker(rank) : nullptr; }
int ifa_engine_worker_plan(ifa_engine *e, int rank, int *out6) { return e && out6 && e->engine.WorkerPlanOf(rank, out6) ? 0 : -1; }

// host-only: the partition rules of the multi-GPU engine (model_loader.cc) for tests and tools
int ifa_partition_slice(int stage, int n_stages, int tp_rank, int tp_size, int layer0, int layer1, int layer, int tensor_id,
                        size_t rows, size_t cols, size_t *out5)
{
    if (!out5) return -1;
    WorkerPlan w; w.stage = stage; w.n_stages = n_stages; w.tp_rank = tp_rank; w.tp_size = tp_size; w.layer0 = layer0; w.layer1 = layer1;
    w.first_stage = stage == 0; w.last_stage = stage + 1 == n_stages;
    TensorSlice sl;
    if (!SliceForWorker(w, layer, tensor_id, rows, cols, sl)) return 0;
    out5[0] = sl.row0; out5[1] = sl.row1; out5[2] = sl.col0; out5[3] = sl.col1; out5[4] = (size_t)sl.local_layer;
    return 1;
}

int ifa_partition_split_layers(int n_layers, int n_groups, int *out_pairs, int capacity_pairs)
{
    if (!out_pairs || n_groups < 1) return -1;
    std::vector<std::pair<int, int>> r;
    SplitGpuLayers(n_layers, n_groups, r);
    for (size_t i = 0; i < r.size() && (int)i < capacity_pairs; i++) { out_pairs[2 * i] = r[i].first; out_pairs[2 * i + 1] = r[i].second; }
    return (int)r.size();
}

// Service shell (host/inferflow_service.*), testable without a device: parses a request body the way the HTTP front does and
// writes the parsed fields back as one JSON object; returns 0, or -1 when the body is rejected / the buffer is too small
int ifa_service_parse_request(const char *body, int is_openai_mode, char *out_json, size_t cap)
{
    if (!body || !out_json || cap == 0) return -1;
    InferFlowRequest r;
    std::string err;
    if (!InferFlowServiceCore::ParseRequest(r, body, is_openai_mode != 0, &err)) { snprintf(out_json, cap, "{\"ret_code\": \"%s\"}", err.c_str()); return -1; }
    std::string ids = "[";
    for (size_t i = 0; i < r.prompt_token_ids.size(); i++) { if (i) ids += ", "; ids += std::to_string(r.prompt_token_ids[i]); }
    ids += "]";
    char tmp[64]; snprintf(tmp, sizeof tmp, "%.4f", r.temperature);
    const std::string js = "{\"prompt_token_ids\": " + ids + ", \"max_output_len\": " + std::to_string(r.max_output_len) + ", \"decoding_alg\": \"" + r.decoding_alg
        + "\", \"random_seed\": " + std::to_string(r.random_seed) + ", \"temperature\": " + tmp + ", \"is_streaming_mode\": " + (r.is_streaming_mode ? "true" : "false")
        + ", \"eos_token_id\": " + std::to_string(r.eos_token_id) + ", \"fn\": \"" + r.fn + "\"}";
    if (js.size() + 1 > cap) return -1;
    memcpy(out_json, js.c_str(), js.size() + 1);
    return 0;
}

// formats a response chunk (native or OpenAI shape) from token ids: the exact strings the HTTP front sends
int ifa_service_format_response(const int *token_ids, int n, int is_end, int is_openai_mode, int is_chunk, int prompt_tokens, char *out_json, size_t cap)
{
    if (!out_json || cap == 0 || n < 0 || (n > 0 && !token_ids)) return -1;
    InferFlowResponseChunk c;
    c.ret_code = "succ"; c.token_ids.assign(token_ids, token_ids + n); c.is_end = is_end != 0; c.prompt_tokens = prompt_tokens;
    std::string js;
    if (is_openai_mode) c.ToJsonOpenAI(js, is_chunk != 0, "ifa-test"); else c.ToJson(js);
    if (js.size() + 1 > cap) return -1;
    memcpy(out_json, js.c_str(), js.size() + 1);
    return 0;
}

// ---- the service LOOP without a device -------------------------------------------------------------------------------------------
// A host-only QueryEngine with InferenceEngine's query-table semantics (AddQuery: > 0 id / 0 busy / -1 too long; Infer: one item
// per query with uncommitted tokens, a query whose context is full is marked ended and gets NO item, a failed step returns false
// without items; Commit appends, is_end ends) over a trivial "model": next token = (last + 1) % vocab.
namespace {
struct LoopbackEngine : QueryEngine {
    struct Q { std::vector<int> tokens; int processed = 0; bool ended = false; };
    std::map<int, Q> qs;
    int max_ctx, max_queries, fail_at, vocab = 1000, next_id = 1, infer_calls = 0;
    LoopbackEngine(int ctx, int mq, int fail) : max_ctx(ctx), max_queries(mq), fail_at(fail) {}
    int AddQuery(const std::vector<int> &t, const QueryOptions &) override {
        if (t.empty() || (int)t.size() >= max_ctx) return -1;
        if ((int)qs.size() >= max_queries) return 0;
        Q q; q.tokens = t; qs[next_id] = q; return next_id++;
    }
    int QueryCount() const override { return (int)qs.size(); }
    bool Infer(InferenceResult &res) override {
        res.items.clear();
        ++infer_calls;
        if (infer_calls == fail_at) return false;                                        // fail_at = N > 0: call N fails once (a retry succeeds)
        if (fail_at < 0 && (infer_calls == -fail_at || infer_calls == -fail_at + 1)) return false;      // fail_at = -N: calls N and N + 1 fail (the retry too)
        for (auto &kv : qs) {
            Q &q = kv.second;
            if (q.ended) continue;
            if ((int)q.tokens.size() >= max_ctx) { q.ended = true; continue; }
            if ((int)q.tokens.size() - q.processed <= 0) continue;
            QueryInferenceResult item; item.query_id = kv.first; item.prefix_len = q.processed;
            IdWeight w; w.id = (q.tokens.back() + 1) % vocab; w.weight = 1.0f;
            item.next_tokens.push_back(w);
            q.processed = (int)q.tokens.size();
            res.items.push_back(item);
        }
        return true;
    }
    bool CommitInferenceResult(const std::map<int, QueryNextToken> &m) override {
        for (const auto &kv : m) { auto it = qs.find(kv.first); if (it == qs.end()) continue; it->second.tokens.push_back(kv.second.id); if (kv.second.is_end) it->second.ended = true; }
        return true;
    }
    bool RemoveQuery(int id) override { return qs.erase(id) != 0; }
    bool QueryEnded(int id) const override { auto it = qs.find(id); return it == qs.end() || it->second.ended; }
    int MaxContextLen() const override { return max_ctx; }
    SamplingStrategyId GetSamplingStrategyId(const std::string &) const override { return SamplingStrategyId::Greedy; }
    std::string Version() const override { return "loopback"; }
    std::string ModelId() const override { return "loopback"; }
    int VocabSize() const override { return vocab; }
};
}

// n_requests queries one after the other through InferFlowServiceCore::ProcessQuery over the engine above; every one must
// RETURN (a handler stuck in its wait loop is reported as "hung": the core is stopped to release it) and hand its slot back
// (an "error.busy" after max_queries earlier requests means a leak).  Writes a JSON list of the results.
int ifa_service_selftest_loop(int max_ctx, int max_queries, int fail_at_infer_call, const int *prompt, int n_prompt, int max_output_len,
                              int eos_token_id, int n_requests, int timeout_ms, char *out_json, size_t cap)
{
    if (!out_json || cap == 0 || n_prompt < 0 || (n_prompt > 0 && !prompt) || n_requests < 1) return -1;
    LoopbackEngine eng(max_ctx, max_queries, fail_at_infer_call);
    InferFlowServiceCore core(eng);
    core.Start();
    std::string js = "[";
    for (int r = 0; r < n_requests; r++) {
        InferFlowRequest req;
        req.prompt_token_ids.assign(prompt, prompt + n_prompt);
        req.max_output_len = max_output_len; req.eos_token_id = eos_token_id;
        InferFlowResponseChunk res;
        bool ok = false, hung = false;
        auto fut = std::async(std::launch::async, [&] { return core.ProcessQuery(res, req, nullptr); });
        if (fut.wait_for(std::chrono::milliseconds(timeout_ms)) != std::future_status::ready) { hung = true; core.Stop(); }
        ok = fut.get();
        if (hung) core.Start();
        std::string ids = "[";
        for (size_t i = 0; i < res.token_ids.size(); i++) { if (i) ids += ", "; ids += std::to_string(res.token_ids[i]); }
        std::string openai; res.ToJsonOpenAI(openai, false, "ifa-test");
        js += std::string(r ? ", " : "") + "{\"ok\": " + (ok ? "true" : "false") + ", \"hung\": " + (hung ? "true" : "false") + ", \"ret_code\": \"" + res.ret_code
            + "\", \"is_end\": " + (res.is_end ? "true" : "false") + ", \"finish_reason\": \"" + res.finish_reason + "\", \"token_ids\": " + ids + "], \"active\": "
            + std::to_string(eng.QueryCount()) + ", \"openai\": " + openai + "}";
    }
    core.Stop();
    js += "]";
    if (js.size() + 1 > cap) return -1;
    memcpy(out_json, js.c_str(), js.size() + 1);
    return 0;
}

} // extern "C"
