// inference_engine.h -- the reference's InferenceEngine surface over the MI355X decode worker.
//
// Mirrors src/transformer/inference_engine.h:32-129 (class InferenceEngine),
// inference_types.h:18-179 (InferenceConfig, QueryNextToken, InferenceResult ...) and
// model.h:72-151 (ModelSpec): same member names, same argument meaning, same error convention
// (bool / query-id returns, a message through LogError -> ifa_engine_last_error(), never throws).
// The host side is plain C++ and reaches the GPU only through the C ABI of include/inferflow_amd.h.
//
// Scope: token-id queries, one model per engine.  `devices` takes the reference's grammar
// (inference_engine.cc:1738-1783): "0" one GPU; "0&1" one device group, tensor parallel (BY_TENSOR); "0;1" two groups,
// consecutive layer ranges (BY_LAYER); "0&1;2&3" HYBRID -- one worker and one host thread per GPU inside this process,
// the exchanges through the C ABI collectives (RCCL).  Tokenizers, prompt templates and the HTTP service are outside
// the hot path.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "sampling_strategy.h"

struct ifa_model;
struct ifa_comm;

namespace inferflow_amd {

struct ModelHyperParams {          // ModelHyperParams, model.h:24-70
    int vocab_size = 0, output_vocab_size = 0;
    int embd_dims = 0, hidden_dim = 0;
    int decoder_layers = 0, decoder_heads = 0, decoder_kv_heads = 0;
    int training_context_len = 0;
    int experts = 0, in_use_experts = 0, moe_top_k = 2;
    bool moe_norm_top_k_prob = true;
};

enum class TensorNormAlg { STD = 0, RMS = 1 };
enum class ActivationFn { SILU = 0, GELU = 1, RELU = 2 };
enum class PositionEmbeddingAlg { EMPTY = 0, ROPE = 1, ALIBI = 2 };

struct ModelSpec {
    std::string sid;
    bool moe_top_k_from_spec = false;      // "moe_top_k" was given in network_structure (config.json does not override it)
    ModelHyperParams hyper_params;
    std::string dir, spec_file, config_file;
    std::vector<std::string> model_files;
    std::string model_file_format;          // "llama2.c" | "safetensors" | "synthetic"
    std::string network_structure = "transformer.llama";
    TensorNormAlg norm_alg = TensorNormAlg::STD;
    ActivationFn activation_fn = ActivationFn::SILU;
    PositionEmbeddingAlg pos_embedding_alg = PositionEmbeddingAlg::ROPE;
    float rope_theta = 10000.0f, partial_rotary_factor = 1.0f, kq_scale = 1.0f;
    float attn_pre_norm_base = 0, ffn_pre_norm_base = 0, output_norm_base = 0;      // RMS weight = base + w (Gemma)
    float attn_out_scale = 1, ffn_out_scale = 1, out_scale = 1;                       // TensorOpr::Scale (MiniCPM)
    bool has_embedding_linear_norm = false;                                           // TensorOpr::LinearNorm on the decoder input (Gemma, MiniCPM)
    float embedding_linear_scale = 0;                                                 // <= 0.0001: sqrt(embd_dims)
    int qk_column_order = 0, qkv_format = 0;
    bool is_parallel_attn = false, mlp_attn_share_input = false;
    bool is_attn_post_as_residual = true;   // model.h:113: with a self_attn.post_norm, the FFN's residual is the normalised tensor
    std::string tensor_name_prefix;
    std::map<std::string, std::string> tensor_name_map;
    std::string decoding_strategy;
    // ids SamplingStrategy::GetSortedTopK never offers (sampling_strategy.cc:281-297): the vocabulary's unk id
    // (StdVocabulary::unk_id_, default 0; -1: none) and Invalid-type tokens
    int unk_token_id = 0;
    std::vector<int> invalid_token_ids;
    std::string decoder_input_template;     // kept for round-tripping the .ini; unused (token-id queries)
    int device_weight_data_type = 1;        // ElementType ids = ifa_dtype; F16
    // device_weight_data_type.<tensor> (inference_engine.cc:1664-1690; tensors attn_wq / attn_wk / attn_wv / attn_wo / ffn_w1 / ffn_w2 /
    // ffn_w3, NetworkStructure::BuildLayerTensorIdMap): per-tensor override of the type above, indexed by IFA_T_* id; -1 = Auto
    int device_weight_data_types[40] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                        -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
    int device_kv_cache_data_type = 8;      // Q8_B32T2 (the reference's default, model.h:137)
    int tensor_quant_threshold = 2000 * 2000;
    static const int DEFAULT_MAX_CONTEXT_LEN = 1024;
    int max_context_len = -1;
    std::vector<std::vector<int>> device_groups;
    // "synthetic" models only: N(0, synthetic_std) weights from seed 1000 + 16*layer + tensor_id
    float synthetic_std = 0.02f;
};

struct InferenceConfig {
    struct DebugOptions { bool is_study_mode = false, show_tensors = false, enable_perf_stat = true; };
    std::vector<ModelSpec> models;
    std::string data_dir;
    int max_concurrent_queries = 10;
    std::vector<std::vector<int>> device_groups;
    int encoder_cpu_layer_count = 0, decoder_cpu_layer_count = 0, cpu_threads = 8;
    std::map<std::string, std::string> prompt_templates;
    bool return_output_tensors = false;
    // extension: queries advancing by one token share ONE batched step (rows GEMM on the matrix cores, five launches per
    // layer) from this many on; below it each runs the fused single-query decode (Llama-2-7B Q4: 2 queries batched give
    // 825 aggregate tok/s against 690 for one after the other)
    int dynamic_batching_min_queries = 2;
    // extension (tests on a 1-GPU box): run a single device through the partition path -- rank thread, communicator of
    // one rank, the C-driven step with its collectives -- instead of the plain single-worker path
    bool force_partition_path = false;
    DebugOptions debug;
};

struct QueryOptions {               // SamplingStrategy::QueryOptions (sampling_strategy.h:75-81)
    int strategy_id = 0;            // SamplingStrategyId: 0 Auto (the model's decoding_strategy, greedy if none), 1 sample.std,
                                    // 2 greedy (device argmax, the hot path), 3 top_k, 4 top_p, 5 fsd, 6 random_fsd, 7 min_p, 8 tfs, 9 typical, 10 mirostat
    int random_seed = 0;            // != 0: seeds the query's generator (reproducible draws)
    float temperature = 1.0f;
    int max_output_len = -1;
};

struct QueryInferenceResult {
    int query_id = 0;
    int prefix_len = 0;
    std::vector<IdWeight> next_tokens;          // [0] = the chosen token (greedy: weight 1; sampled: its pool probability)
    std::vector<uint16_t> output_tensor;        // F16 logits [output_rows][output_cols] if return_output_tensors
    int output_rows = 0, output_cols = 0;
};

struct QueryNextToken { int id = 0; bool is_end = false; };

struct InferencePerfStat { std::map<uint32_t, float> time_map; };   // key 0: the step end to end (ms); study mode: the reference's per-phase keys, (layer + 1) * 10000 + phase

struct InferenceResult {
    std::vector<QueryInferenceResult> items;
    InferencePerfStat perf_stat;
};

// What the service shell (inferflow_service.h) needs from an engine: the query-level calls of the reference's InferenceEngine
// (src/transformer/inference_engine.h:41-75) plus two facts the reference's service reads off its query table -- whether a
// query has ended inside the engine (context full) and the context limit.  InferenceEngine implements it; the CPU tests drive
// the service loop over a host-only implementation.
class QueryEngine {
public:
    virtual ~QueryEngine() {}
    virtual int AddQuery(const std::vector<int> &tokens, const QueryOptions &query_options) = 0;
    virtual int QueryCount() const = 0;
    virtual bool Infer(InferenceResult &res) = 0;
    virtual bool CommitInferenceResult(const std::map<int, QueryNextToken> &query_map) = 0;
    virtual bool RemoveQuery(int query_id) = 0;
    virtual bool QueryEnded(int query_id) const = 0;       // true also for an unknown id
    virtual int MaxContextLen() const = 0;
    virtual SamplingStrategyId GetSamplingStrategyId(const std::string &str = "") const = 0;
    virtual std::string Version() const = 0;
    virtual std::string ModelId() const = 0;
    virtual int VocabSize() const = 0;
};

class InferenceEngine : public QueryEngine {
public:
    InferenceEngine();
    ~InferenceEngine() override;
    InferenceEngine(const InferenceEngine &) = delete;
    InferenceEngine &operator=(const InferenceEngine &) = delete;
    void Clear();

    static bool LoadConfig(InferenceConfig &config, const std::string &config_path,
                           const std::string &section, const std::string &data_root_dir = "");
    bool Init(const InferenceConfig &cfg);

    // > 0: query id, 0: busy (max_concurrent_queries reached), < 0: error
    int AddQuery(const std::vector<int> &tokens, const QueryOptions &query_options) override;
    int QueryCount() const override;
    // one step for every active query: prefill of the pending tokens, or one decode step.  A query whose context is full
    // (tokens == max_context_len) is marked ended and gets NO item (QueryEnded tells the caller).
    bool Infer(InferenceResult &res) override;
    bool CommitInferenceResult(const std::map<int, QueryNextToken> &query_map) override;
    bool RemoveQuery(int query_id) override;
    bool QueryEnded(int query_id) const override;
    int MaxContextLen() const override { return spec_.max_context_len > 0 ? spec_.max_context_len : ModelSpec::DEFAULT_MAX_CONTEXT_LEN; }
    std::string ModelId() const override { return spec_.sid; }
    int VocabSize() const override { return spec_.hyper_params.vocab_size; }

    // Extension: n greedy steps with the token fed back on the device (hipGraph replay, no host
    // round trip per token).  Equivalent to n x {Infer, CommitInferenceResult(greedy)}.
    bool Generate(int query_id, int n_steps, std::vector<int> &new_tokens, float *gpu_ms = nullptr);

    // id of a strategy name ("sample.top_p" ...); empty: the model's own decoding_strategy
    SamplingStrategyId GetSamplingStrategyId(const std::string &str = "") const override;
    const ModelSpec &model_spec() const { return spec_; }
    std::string Version() const override { return "inferflow_amd 0.1 (MI355X)"; }
    const InferenceConfig &config() const { return config_; }
    // "0 (E2E)\t<ms>" then "<key>\t<ms>" per line, ascending keys (InferenceEngine::PrintPerfStat, inference_engine.cc:2108-2120)
    static void PrintPerfStat(FILE *strm, const InferencePerfStat &perf_stat)
    {
        for (const auto &kv : perf_stat.time_map) {
            if (kv.first == 0) fprintf(strm, "0 (E2E)\t%g\n", kv.second);
            else fprintf(strm, "%u\t%g\n", kv.first, kv.second);
        }
    }
    int default_device_id() const { return device_; }
    int PartitionRanks() const;     // workers of the multi-GPU partition (1: single device)
    ifa_model *worker() { return model_; }
    // worker of partition rank r and its place in the partition (stage, n_stages, tp_rank, tp_size, layer0, layer1); rank 0 of a
    // single-device engine is worker().  The tests read the ranks' weight slices back and rebuild the whole model for the oracle.
    ifa_model *worker(int rank);
    bool WorkerPlanOf(int rank, int out6[6]) const;

private:
    struct Query {
        int id = 0;
        std::vector<int> tokens;    // committed tokens (prompt + generated)
        int processed = 0;          // tokens whose KV rows are in the cache
        QueryOptions options;
        bool ended = false;
        int kv_slot = 0;            // this query's KV cache inside the worker (ifa_model_select_kv)
        SamplingStrategyId strategy = SamplingStrategyId::Greedy;
        StdSamplingConfig sampling; // per query copy, like StdQueryData::config
        JavaRandom rng;
        SamplingState sampling_state;   // Mirostat's mu, the FSD n-gram model, the EOS bypass count
    };
    bool SampleRow(Query &q, const uint16_t *logits_row, QueryInferenceResult &item);
    // ---- multi-GPU partitions (devices = 0&1 | 0;1 | 0&1;2&3): one worker and one host thread per GPU, like the
    // reference's Infer_TensorParallelism / Infer_Std over GpuInferenceWorker threads (inference_engine.cc:1161-1296)
    struct MultiGpu;
    MultiGpu *multi_ = nullptr;
    bool InitMulti(const std::vector<std::vector<int>> &groups);
    bool MultiBatchStep(const std::vector<int> &toks, const std::vector<int> &pos, const std::vector<int> &slots, std::vector<int> &next,
                        bool want_tensor, std::vector<uint16_t> &all);
    bool MultiStep(Query &q, int n_new, bool want_tensor, QueryInferenceResult &item, int &next);
    InferenceConfig config_;
    ModelSpec spec_;
    ifa_model *model_ = nullptr;
    int device_ = 0;
    int next_query_id_ = 1;
    int kv_slots_ = 1;
    SamplingStrategyId default_strategy_ = SamplingStrategyId::Greedy;
    StdSamplingConfig default_sampling_;
    bool perf_phases_ = false;      // study mode: the per-phase keys of InferencePerfStat from the worker (ifa_model_perf_stat)
    bool host_greedy_ = false;      // more excluded token ids than the device argmax holds: greedy selection on the host
    std::map<int, Query> queries_;
    void *logits_dev_ = nullptr;
    size_t logits_rows_ = 0;
};

// error text of the last failed call on this thread (the reference logs through LogError)
const char *EngineLastError();
void EngineSetError(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

// model loading (model_loader.cc)
bool LoadModelSpecJson(ModelSpec &spec, const std::string &path);
bool BuildWorker(ifa_model **out, ModelSpec &spec, int device);

// One worker of a multi-GPU partition (MultiGpuStrategy BY_LAYER / BY_TENSOR / HYBRID, src/transformer/model.h:61-66;
// "devices = 0&1;2&3": groups separated by ';' hold consecutive layer ranges, devices joined by '&' share every layer).
struct WorkerPlan {
    int device = 0;
    int stage = 0, n_stages = 1;        // device group (layer range) of this worker
    int tp_rank = 0, tp_size = 1;       // position inside the group
    int layer0 = 0, layer1 = -1;        // global layers [layer0, layer1); layer1 < 0: all (filled in by the loader)
    bool first_stage = true, last_stage = true;
    ifa_model *model = nullptr;
};
struct TensorSlice { size_t row0, row1, col0, col1; int local_layer; };
// the slice of tensor (layer, tid) [rows][cols] worker w holds; false: none of it
bool SliceForWorker(const WorkerPlan &w, int layer, int tid, size_t rows, size_t cols, TensorSlice &sl);
void SplitGpuLayers(int n_layers, int n_groups, std::vector<std::pair<int, int>> &ranges);
// plans: one entry per (group, rank) in the reference's device order; layer ranges are assigned from the checkpoint's layer count
bool BuildWorkers(std::vector<WorkerPlan> &plans, ModelSpec &spec);

} // namespace inferflow_amd
