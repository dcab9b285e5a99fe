/* inferflow_engine.h -- C ABI over the C++ InferenceEngine facade (inferflow_amd/host/inference_engine.h).
 *
 * What a non-C++ caller (ctypes, cgo, JNI ...) binds to drive the reference's serving loop
 *   LoadConfig -> Init -> AddQuery -> { Infer -> CommitInferenceResult }* -> RemoveQuery
 * (InferenceEngine, src/transformer/inference_engine.h:32-129; driver loop src/tools/llm_inference.cc:345-457).
 * Return conventions are the reference's: 0/false = failure with the text in ifa_engine_last_error(),
 * AddQuery: > 0 query id, 0 busy, < 0 error.
 */
#ifndef INFERFLOW_ENGINE_H
#define INFERFLOW_ENGINE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ifa_engine ifa_engine;

/* InferenceEngine::LoadConfig(config_path, section, data_root_dir) + Init(); NULL on failure */
ifa_engine *ifa_engine_create(const char *ini_path, const char *section, const char *data_root_dir);
void ifa_engine_destroy(ifa_engine *e);
const char *ifa_engine_last_error(void);

/* AddQuery(tokens, QueryOptions{strategy greedy}) */
int ifa_engine_add_query(ifa_engine *e, const int *tokens, int n_tokens);
/* AddQuery with SamplingStrategy::QueryOptions: strategy_id = SamplingStrategyId (0 Auto = the model's decoding_strategy or
 * greedy, 1 sample.std, 2 greedy, 3 top_k, 4 top_p, 5 fsd, 6 random_fsd, 7 min_p, 8 tfs, 9 typical, 10 mirostat), random_seed != 0 seeds the
 * query's generator (sslib Random = the java.util.Random LCG), temperature as in SamplingStrategy::SoftMax */
int ifa_engine_add_query_ex(ifa_engine *e, const int *tokens, int n_tokens, int strategy_id, int random_seed, float temperature);
/* GetSamplingStrategyId(name): "sample.top_p", "greedy", ...; NULL/"" = the loaded model's default; 0 if unknown */
int ifa_engine_strategy_id(ifa_engine *e, const char *name);
/* host-only (no GPU): StdSamplingStrategy::ChooseTokens (src/transformer/sampling_strategy.cc:359-431) on one F16 logits
 * row, n_draws consecutive draws from a generator seeded with `seed`; writes the drawn ids / pool probabilities and the
 * pool after the top_p / max_k cut; returns the pool size or -1 */
int ifa_sampling_choose(const uint16_t *logits_f16, int vocab, int strategy_id, int max_k, float top_p, int pool_size,
                        float temperature, long long seed, int n_draws, int *out_ids, float *out_probs,
                        int *pool_ids, float *pool_probs, int pool_capacity);
/* the same for every strategy (adds 5 fsd, 6 random_fsd -- text_tokens = the query's tokens so far, consecutive draws extend the
 * n-gram model with the drawn tokens -- 7 min_p, 8 tfs, 9 typical, 10 mirostat): params9 = {max_k, top_p, pool_size, min_p, tfs z,
 * typical p, mirostat eta, mirostat tau, eos_bypassing_max}; *mirostat_mu_inout (nullable; NaN = unset -> 2 tau) carries mu across calls */
int ifa_sampling_choose_ex(const uint16_t *logits_f16, int vocab, int strategy_id, const float *params9, float temperature,
                           long long seed, int n_draws, int *out_ids, float *out_probs, int *pool_ids, float *pool_probs,
                           int pool_capacity, float *mirostat_mu_inout, const int *text_tokens, int n_text);
/* the first n NextDouble() values of the generator seeded with `seed` (known-answer tests of the LCG) */
int ifa_sampling_random_doubles(long long seed, int n, double *out);
int ifa_engine_query_count(ifa_engine *e);
int ifa_engine_remove_query(ifa_engine *e, int query_id);           /* 1 removed, 0 unknown id */

/* Infer(): one step for every active query.  Writes up to `capacity` (query id, greedy next token) pairs and
 * returns how many were produced, or -1.  With return_output_tensors = true the F16 logits of the most recent
 * step of a query can be fetched with ifa_engine_last_logits. */
int ifa_engine_infer(ifa_engine *e, int *query_ids, int *next_tokens, int capacity);
/* CommitInferenceResult({query_id: {token, is_end}}) */
int ifa_engine_commit(ifa_engine *e, const int *query_ids, const int *tokens, const int *is_end, int n);
/* rows/cols of the logits kept from the last Infer() for this query; copies min(capacity, rows*cols) halfs */
int ifa_engine_last_logits(ifa_engine *e, int query_id, uint16_t *dst_f16, size_t capacity, int *rows, int *cols);

/* InferenceResult::perf_stat of the last Infer() (InferencePerfStat::time_map, src/transformer/inference_types.h): up to `capacity`
 * (key, milliseconds) pairs in ascending key order; returns how many keys there are.  Key 0 = the step end to end
 * (inference_engine.cc:986-988); with is_study_mode = true in the .ini also the per-phase keys (layer + 1) * 10000 + phase of
 * GpuInferenceWorker::UpdatePerfStat (inference_worker.cc:2670-2697; measured on the op-by-op step, see ifa_model_perf_stat). */
int ifa_engine_perf_stat(ifa_engine *e, unsigned *keys, float *ms, int capacity);

/* extension: n greedy steps with device-side token feedback (graph replay); returns tokens written or -1 */
int ifa_engine_generate(ifa_engine *e, int query_id, int n_steps, int *out_tokens, float *gpu_ms);

/* the reference's perplexity harness (src/tools/perplexity.cc:41-284) over a token-id stream: windows of max_length
 * every `stride` tokens, each scored from its whole-prompt logits; needs return_output_tensors = true in the .ini and
 * no active query.  1 ok (PPL, its error estimate, scored-token count), 0 failure. */
int ifa_engine_perplexity(ifa_engine *e, const int *tokens, int n_tokens, int max_length, int stride,
                          double *ppl, double *ppl_stderr, long long *count);

/* host-only: -log softmax(logits)[token_id] of one F16 logits row with the tool's arithmetic (perplexity.cc:100-119); < 0 on bad arguments */
double ifa_perplexity_token_nll(const uint16_t *logits_f16, int vocab, int token_id);

/* facts of the loaded model: "vocab_size", "embd_dims", "hidden_dim", "decoder_layers", "decoder_heads",
 * "decoder_kv_heads", "max_context_len", "device_weight_data_type", "device_kv_cache_data_type", "partition_ranks"
 * (workers of the multi-GPU partition; 1 = single device); -1 if unknown */
int ifa_engine_model_info(ifa_engine *e, const char *key);

/* the per-device worker of partition rank `rank` (an ifa_model * for the ifa_model_* calls of inferflow_amd.h; rank 0 of a
 * single-device engine) and its place in the partition {stage, n_stages, tp_rank, tp_size, layer0, layer1}: the counterpart of
 * reaching a GpuInferenceWorker through InferenceEngine (src/transformer/inference_engine.cc:1916-1984).  NULL / -1: no such rank. */
void *ifa_engine_worker(ifa_engine *e, int rank);
int ifa_engine_worker_plan(ifa_engine *e, int rank, int *out6);

/* host-only: the partition rules the engine applies to "devices = 0&1;2&3" (BY_TENSOR slices of
 * network_builder.cc:1594-1686 / device_tensor_builder.cu:203-239, layer ranges of NetworkBuilder::SplitGpuLayers
 * :2094-2118).  slice: 1 + {row0, row1, col0, col1, local layer} of tensor (layer, tensor_id) [rows][cols] for the worker
 * at (stage, tp_rank), 0 if it holds none of it, -1 on bad arguments.  split_layers: number of groups written as
 * (start, end) pairs. */
int ifa_partition_slice(int stage, int n_stages, int tp_rank, int tp_size, int layer0, int layer1, int layer, int tensor_id,
                        size_t rows, size_t cols, size_t *out5);
int ifa_partition_split_layers(int n_layers, int n_groups, int *out_pairs, int capacity_pairs);

/* ---- service shell (host/inferflow_service.*: the token-id counterpart of src/service/inferflow_service.cc; the process is
 * bin/ifa_service <config.ini> [--port N]).  These two host-only entry points expose its request parser and response
 * formatter -- native shape and the OpenAI-shaped /chat/completions one -- so that they can be tested without a device:
 * parse writes the parsed fields back as one JSON object; both return 0, or -1 (rejected body / buffer too small). */
int ifa_service_parse_request(const char *body, int is_openai_mode, char *out_json, size_t cap);
int ifa_service_format_response(const int *token_ids, int n, int is_end, int is_openai_mode, int is_chunk, int prompt_tokens,
                                char *out_json, size_t cap);
/* host-only: the service's Infer / Commit LOOP (InferFlowServiceCore, the counterpart of src/service/inferflow_service.cc:60-129)
 * over a loopback engine with InferenceEngine's query-table semantics (a query whose context is full is ended WITHOUT an item; the
 * fail_at_infer_call-th Infer returns false).  Runs n_requests queries one after the other; writes a JSON list of
 * {ok, hung, ret_code, is_end, finish_reason, token_ids, active, openai}.  0, or -1 on bad arguments / a small buffer. */
int ifa_service_selftest_loop(int max_ctx, int max_queries, int fail_at_infer_call, const int *prompt, int n_prompt, int max_output_len,
                              int eos_token_id, int n_requests, int timeout_ms, char *out_json, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
