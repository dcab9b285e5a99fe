// inference_engine.cc -- InferenceEngine over the MI355X decode worker (see inference_engine.h).
// Step semantics follow InferenceEngine::Infer_Std (src/transformer/inference_engine.cc:1161-1220):
// every Infer() advances each active query by one step -- the whole pending prompt on the first
// step (prefill), one token afterwards -- and reports the candidates of the next token; the caller
// picks one and commits it with CommitInferenceResult (llm_inference.cc:345-457).
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "inferflow_amd.h"
#include "inference_engine.h"
#include "ifa_ini.h"

namespace inferflow_amd {

// ---------------------------------------------------------------------------------------------- multi-GPU partitions
// One worker (ifa_model) and one persistent host thread per GPU.  The reference creates and joins a thread per GPU inside
// every Infer() (inference_engine.cc:1203-1206, 1261-1283) and lets the workers rendezvous through GpuInfGlobalData's
// mutex; here the threads live as long as the engine and every exchange is a collective of the C ABI enqueued on the
// worker's stream (csrc/ifa_comm.hip), so a thread only ever blocks at the end of its step.
struct InferenceEngine::MultiGpu {
    std::vector<WorkerPlan> plans;
    std::vector<ifa_comm *> world, tp;          // per rank (world: only with several device groups; tp: only with groups of > 1)
    std::vector<ifa_tp_topology> topo;
    std::vector<void *> shard_dev;              // per rank: logits shard buffer [rows][V / P] (last group only)
    size_t shard_rows = 0;
    int G = 1, P = 1;
    bool force_collectives = false;
    // thread pool
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<int(int)> job;
    uint64_t generation = 0;
    int pending = 0;
    bool stop = false;
    int first_failed = -1;
    bool broken = false;      // a rank failed inside a step: the communicators were aborted, the engine cannot continue
    std::vector<std::string> errors;

    // A rank that fails before or between the collectives of a step leaves its peers blocked in theirs (RCCL, or the
    // loopback group's rendezvous) and Run() would never return.  The failing rank's thread aborts every communicator of
    // the job: the peers come back with an error, Run() reports the FIRST failure.
    void AbortGroups()
    {
        for (ifa_comm *c : tp) if (c) ifa_comm_abort(c);
        for (ifa_comm *c : world) if (c) ifa_comm_abort(c);
    }

    void Start()
    {
        const int n = (int)plans.size();
        errors.assign((size_t)n, std::string());
        for (int r = 0; r < n; r++)
            threads.emplace_back([this, r]() {
                uint64_t seen = 0;
                for (;;) {
                    std::function<int(int)> fn;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_job.wait(lk, [&] { return stop || generation != seen; });
                        if (stop) return;
                        seen = generation; fn = job;
                    }
                    const int rc = fn(r);
                    bool first_failure = false;
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        errors[(size_t)r] = rc == 0 ? std::string() : std::string(ifa_last_error());   // (thread-local message)
                        if (rc != 0 && errors[(size_t)r].empty()) errors[(size_t)r] = "error " + std::to_string(rc);
                        if (rc != 0 && !broken) { broken = true; first_failure = true; first_failed = r; }
                    }
                    if (first_failure && plans.size() > 1) AbortGroups();
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    // fn(rank) on every rank's thread at once; false + message if any failed
    bool Run(const std::function<int(int)> &fn, const char *what)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (broken) { EngineSetError("%s: the engine's device group was aborted after an earlier failure; create a new engine", what); return false; }
            job = fn; pending = (int)plans.size(); generation++;
        }
        cv_job.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        if (first_failed >= 0) {      // the rank whose failure started it (the others only report the abort)
            const size_t r = (size_t)first_failed;
            EngineSetError("%s failed on rank %zu (device %d): %s", what, r, plans[r].device, errors[r].c_str());
            return false;
        }
        for (size_t r = 0; r < errors.size(); r++)
            if (!errors[r].empty()) { EngineSetError("%s failed on rank %zu (device %d): %s", what, r, plans[r].device, errors[r].c_str()); return false; }
        return true;
    }
    ~MultiGpu()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_job.notify_all();
        for (std::thread &t : threads) if (t.joinable()) t.join();
        for (size_t r = 0; r < plans.size(); r++) {
            if (r < shard_dev.size() && shard_dev[r]) { ifa_set_device(plans[r].device); ifa_free(shard_dev[r]); }
            if (plans[r].model) ifa_model_destroy(plans[r].model);
        }
        for (ifa_comm *c : tp) if (c) ifa_comm_destroy(c);
        for (ifa_comm *c : world) if (c) ifa_comm_destroy(c);
    }
};

InferenceEngine::InferenceEngine() {}
InferenceEngine::~InferenceEngine() { Clear(); }

void InferenceEngine::Clear()
{
    if (multi_) { delete multi_; multi_ = nullptr; model_ = nullptr; }
    if (model_) { ifa_model_destroy(model_); model_ = nullptr; }
    if (logits_dev_) { ifa_free(logits_dev_); logits_dev_ = nullptr; logits_rows_ = 0; }
    queries_.clear();
}

static bool LoadDeviceGroups(std::vector<std::vector<int>> &groups, const IniConfig &cfg, const std::string &section, const std::string &key)
{
    // "0;1" = two groups (by layer), "0&1" = one group of two (by tensor)  (inference_engine.cc:1738-1783)
    groups.clear();
    std::string str;
    cfg.GetItem(section, key, str);
    for (std::string tok : IniConfig::Split(str, ",;")) {
        tok = IniConfig::Trim(tok);
        if (tok.empty()) continue;
        std::vector<int> sub;
        for (std::string s : IniConfig::Split(tok, "&|")) { s = IniConfig::Trim(s); if (!s.empty()) sub.push_back(atoi(s.c_str())); }
        groups.push_back(sub);
    }
    for (size_t g = 1; g < groups.size(); g++)
        if (groups[g].size() != groups[0].size()) {
            EngineSetError("All device groups should have the same size: %zu vs. %zu", groups[0].size(), groups[g].size());
            return false;
        }
    return true;
}

static bool LoadModelSpec(ModelSpec &spec, const IniConfig &cfg, const std::string &section)
{
    if (!cfg.GetItem(section, "model_dir", spec.dir) || spec.dir.empty()) {
        EngineSetError("The directory of model \"%s\" should not be empty", spec.sid.c_str()); return false;
    }
    if (spec.dir.back() != '/' && spec.dir.back() != '\\') spec.dir += '/';
    if (!cfg.GetItem(section, "model_specification_file", spec.spec_file)) cfg.GetItem(section, "model_spec_file", spec.spec_file);
    if (spec.spec_file.empty()) { EngineSetError("The specification file of model \"%s\" should not be empty", spec.sid.c_str()); return false; }
    cfg.GetItem(section, "decoding_strategy", spec.decoding_strategy);
    cfg.GetItem(section, "decoder_input_template", spec.decoder_input_template);
    cfg.GetItem(section, "prompt_template", spec.decoder_input_template);
    std::string str;
    if (cfg.GetItem(section, "device_weight_data_type", str) && !str.empty()) {
        const int dt = ifa_dtype_from_name(IniConfig::Lower(str).c_str());
        if (dt < 0) { EngineSetError("Invalid device_weight_data_type for model %s", spec.sid.c_str()); return false; }
        spec.device_weight_data_type = dt;
    }
    {   // device_weight_data_type.<tensor>: element size >= 2 -> F16 (inference_engine.cc:1685-1687)
        static const struct { const char *name; int tid; } kTensors[] = {{"attn_wq", IFA_T_WQ}, {"attn_wk", IFA_T_WK}, {"attn_wv", IFA_T_WV},
            {"attn_wo", IFA_T_WO}, {"ffn_w1", IFA_T_W1}, {"ffn_w2", IFA_T_W2}, {"ffn_w3", IFA_T_W3}};
        for (const auto &kt : kTensors) {
            std::string v;
            if (!cfg.GetItem(section, std::string("device_weight_data_type.") + kt.name, v) || v.empty()) continue;
            const int dt = ifa_dtype_from_name(IniConfig::Lower(v).c_str());
            if (dt < 0) { EngineSetError("Invalid device_weight_data_type.%s for model %s", kt.name, spec.sid.c_str()); return false; }
            spec.device_weight_data_types[kt.tid] = (dt == IFA_F32 || dt == IFA_F16) ? IFA_F16 : dt;
        }
    }
    str.clear();
    if (cfg.GetItem(section, "device_kv_cache_data_type", str) && !str.empty()) {
        const int dt = ifa_dtype_from_name(IniConfig::Lower(str).c_str());
        if (dt < 0) { EngineSetError("Invalid device_kv_cache_data_type for model %s", spec.sid.c_str()); return false; }
        // element size >= 2 -> F16, anything smaller -> Q8_B32T2   (inference_engine.cc:1701-1703)
        spec.device_kv_cache_data_type = (dt == IFA_F32 || dt == IFA_F16) ? IFA_F16 : IFA_Q8_B32T2;
    }
    cfg.GetItem(section, "tensor_quant_threshold", spec.tensor_quant_threshold);
    if (!LoadDeviceGroups(spec.device_groups, cfg, section, "devices")) return false;
    cfg.GetItem(section, "max_context_len", spec.max_context_len);
    const bool is_abs = !spec.spec_file.empty() && spec.spec_file[0] == '/';
    return LoadModelSpecJson(spec, is_abs ? spec.spec_file : spec.dir + spec.spec_file);
}

bool InferenceEngine::LoadConfig(InferenceConfig &config, const std::string &config_path,
                                 const std::string &section, const std::string &data_root_dir)
{
    IniConfig cfg; std::string err;
    if (!cfg.Load(config_path, &err)) { EngineSetError("Failed to load the configuration file: %s", err.c_str()); return false; }
    std::string root = data_root_dir;
    if (root.empty()) { IniConfig probe; probe.Load(config_path); probe.GetItem("app_env.base", "data_root_dir", root); }
    if (!root.empty()) cfg.AddMacro("data_root_dir", root);
    config.data_dir = root;
    std::string global_model_dir;
    cfg.GetItem("main", "global_model_dir", global_model_dir);
    cfg.AddMacro("global_model_dir", global_model_dir);
    if (!cfg.HasSection(section)) { EngineSetError("Section [%s] is missing in %s", section.c_str(), config_path.c_str()); return false; }
    if (!LoadDeviceGroups(config.device_groups, cfg, section, "devices")) return false;
    if (config.device_groups.empty()) config.device_groups.push_back({0});
    int cpu_layers = 0;
    if (cfg.GetItem(section, "cpu_layer_count", cpu_layers)) config.decoder_cpu_layer_count = cpu_layers;
    cfg.GetItem(section, "encoder_cpu_layer_count", config.encoder_cpu_layer_count);
    cfg.GetItem(section, "decoder_cpu_layer_count", config.decoder_cpu_layer_count);
    std::string models;
    if (!cfg.GetItem(section, "models", models) || IniConfig::Trim(models).empty()) {
        EngineSetError("Item \"models\" is missing in section [%s]", section.c_str()); return false;
    }
    config.models.clear();
    for (std::string name : IniConfig::Split(models, ",;")) {
        name = IniConfig::Trim(name);
        if (name.empty()) continue;
        ModelSpec spec; spec.sid = name;
        cfg.AddMacro("model_name", name);
        if (!LoadModelSpec(spec, cfg, "model." + name)) return false;
        if (spec.device_groups.empty()) spec.device_groups = config.device_groups;
        config.models.push_back(spec);
    }
    cfg.GetItem(section, "max_concurrent_queries", config.max_concurrent_queries);
    cfg.GetItem(section, "cpu_threads", config.cpu_threads);
    cfg.GetItem(section, "return_output_tensors", config.return_output_tensors);
    cfg.GetItem(section, "dynamic_batching_min_queries", config.dynamic_batching_min_queries);
    cfg.GetItem(section, "force_partition_path", config.force_partition_path);
    cfg.GetItem(section, "is_study_mode", config.debug.is_study_mode);
    cfg.GetItem(section, "show_tensors", config.debug.show_tensors);
    return true;
}

bool InferenceEngine::Init(const InferenceConfig &cfg)
{
    Clear();
    config_ = cfg;
    if (cfg.models.empty()) { EngineSetError("No model is configured"); return false; }
    if (cfg.models.size() > 1) { EngineSetError("One model per engine (got %zu)", cfg.models.size()); return false; }
    spec_ = cfg.models[0];
    if (cfg.decoder_cpu_layer_count > 0) { EngineSetError("decoder_cpu_layer_count > 0: CPU layers are outside this engine"); return false; }
    const auto &groups = spec_.device_groups.empty() ? cfg.device_groups : spec_.device_groups;
    const bool multi = groups.size() > 1 || (!groups.empty() && groups[0].size() > 1) || cfg.force_partition_path;
    device_ = groups.empty() || groups[0].empty() ? 0 : groups[0][0];
    for (const auto &g : groups)
        for (int d : g)
            if (d < 0 || d >= ifa_device_count()) { EngineSetError("device %d is not available (%d visible)", d, ifa_device_count()); return false; }
    if (device_ < 0 || device_ >= ifa_device_count()) { EngineSetError("device %d is not available (%d visible)", device_, ifa_device_count()); return false; }
    default_strategy_ = SamplingStrategyId::Greedy; default_sampling_ = StdSamplingConfig();
    if (!spec_.decoding_strategy.empty()) {
        SamplingStrategyId sid; std::string err;
        if (!ParseDecodingStrategy(spec_.decoding_strategy, sid, default_sampling_, &err)) { EngineSetError("%s for model %s", err.c_str(), spec_.sid.c_str()); return false; }
        if (sid != SamplingStrategyId::Auto) default_strategy_ = sid;
        if (!IsSupportedStrategy(default_strategy_)) { EngineSetError("decoding_strategy \"%s\" of model %s is not supported", spec_.decoding_strategy.c_str(), spec_.sid.c_str()); return false; }
    }
    if (multi) {
        std::vector<std::vector<int>> gs = groups;
        if (gs.empty()) gs.push_back({device_});
        if (!InitMulti(gs)) { Clear(); return false; }
    } else if (!BuildWorker(&model_, spec_, device_)) return false;
    {   // the ids greedy / sampled selection never offers (GetSortedTopK): device argmax and host pool alike
        std::vector<int> excl;
        if (spec_.unk_token_id >= 0 && spec_.unk_token_id < spec_.hyper_params.vocab_size) excl.push_back(spec_.unk_token_id);
        for (int id : spec_.invalid_token_ids)
            if (id >= 0 && id < spec_.hyper_params.vocab_size && std::find(excl.begin(), excl.end(), id) == excl.end()) excl.push_back(id);
        // GetSortedTopK skips EVERY Invalid-type token: the host pool keeps the full list.  The device argmax holds three
        // ids; a vocabulary with more makes greedy queries select on the host too (one logits row per step), so that no
        // id is silently dropped
        default_sampling_.excluded_ids = excl;
        host_greedy_ = excl.size() > 3;
        if (excl.size() > 3) excl.resize(3);
        std::vector<ifa_model *> all;
        if (multi_) for (WorkerPlan &w : multi_->plans) all.push_back(w.model); else all.push_back(model_);
        for (ifa_model *mm : all)
            if (ifa_model_set_excluded_tokens(mm, excl.data(), (int)excl.size()) != IFA_OK) { EngineSetError("excluded tokens: %s", ifa_last_error()); Clear(); return false; }
    }
    // Per-phase keys of InferencePerfStat ((layer + 1) * 10000 + phase, inference_worker.cc:2670-2697) are filled in study mode only: the
    // reference times the host side of its launches for free, here the phases exist as separate launches only on the op-by-op step
    // (worker option perf_stat).  Key 0 (end to end) is filled whenever enable_perf_stat is on, like inference_engine.cc:986-988.
    perf_phases_ = !multi_ && config_.debug.is_study_mode && config_.debug.enable_perf_stat;
    if (perf_phases_ && ifa_model_set_option(model_, "perf_stat", 1) != IFA_OK) { EngineSetError("perf_stat: %s", ifa_last_error()); Clear(); return false; }
    // one KV cache per concurrent query, like the reference's per-query LayerKVCache sets
    kv_slots_ = std::max(1, std::min(config_.max_concurrent_queries, 64));
    {
        std::vector<ifa_model *> all;
        if (multi_) for (WorkerPlan &w : multi_->plans) all.push_back(w.model); else all.push_back(model_);
        for (ifa_model *mm : all)
            if (ifa_model_kv_slots(mm, kv_slots_) != IFA_OK) { EngineSetError("KV caches for %d queries: %s", kv_slots_, ifa_last_error()); Clear(); return false; }
    }
    return true;
}

// devices = G groups of P: workers in the reference's order (rank = group * P + position), layer ranges by
// SplitGpuLayers, BY_TENSOR slices inside a group (model_loader.cc), one communicator per group + one for the job
bool InferenceEngine::InitMulti(const std::vector<std::vector<int>> &groups)
{
    const int G = (int)groups.size(), P = (int)groups[0].size();
    if (P < 1) { EngineSetError("empty device group"); return false; }
    multi_ = new MultiGpu();
    MultiGpu &M = *multi_;
    M.G = G; M.P = P; M.force_collectives = config_.force_partition_path;
    std::vector<int> all_devices;
    for (int g = 0; g < G; g++)
        for (int r = 0; r < P; r++) {
            WorkerPlan w;
            w.device = groups[(size_t)g][(size_t)r]; w.stage = g; w.n_stages = G; w.tp_rank = r; w.tp_size = P;
            M.plans.push_back(w);
            all_devices.push_back(w.device);
        }
    // a device named more than once: only as "every rank on ONE device" (loopback groups of the C ABI: the multi-rank
    // paths on a 1-GPU box, tests); anything else is a configuration mistake
    bool dup = false, all_same = true;
    for (size_t i = 0; i < all_devices.size(); i++) {
        all_same = all_same && all_devices[i] == all_devices[0];
        for (size_t j = i + 1; j < all_devices.size(); j++) dup = dup || all_devices[i] == all_devices[j];
    }
    if (dup && !all_same) { EngineSetError("a device appears twice in `devices`"); return false; }
    if (!BuildWorkers(M.plans, spec_)) return false;
    const int R = G * P;
    M.world.assign((size_t)R, nullptr); M.tp.assign((size_t)R, nullptr);
    if (G > 1 && ifa_comm_init_all(all_devices.data(), R, M.world.data()) != IFA_OK) { EngineSetError("job communicator: %s", ifa_last_error()); return false; }
    if (P > 1 || M.force_collectives)
        for (int g = 0; g < G; g++)
            if (ifa_comm_init_all(groups[(size_t)g].data(), P, M.tp.data() + (size_t)g * P) != IFA_OK) { EngineSetError("group communicator: %s", ifa_last_error()); return false; }
    const int V = spec_.hyper_params.vocab_size;
    M.topo.resize((size_t)R);
    for (int i = 0; i < R; i++) {
        ifa_tp_topology &t = M.topo[(size_t)i];
        memset(&t, 0, sizeof(t));
        const int g = i / P, r = i % P;
        t.tp = M.tp[(size_t)i]; t.world = M.world[(size_t)i];
        t.stage = g; t.n_stages = G;
        t.prev_rank = g > 0 ? i - P : -1; t.next_rank = g + 1 < G ? i + P : -1;
        t.token_src = (G - 1) * P;                  // first rank of the last group announces the token
        t.vocab_offset = r * (V / P);
        t.force_collectives = M.force_collectives ? 1 : 0;
    }
    M.shard_dev.assign((size_t)R, nullptr);
    M.Start();
    model_ = M.plans[0].model;      // (handle for model_info-style queries; steps go through the rank threads)
    return true;
}

// one step of one query on every rank: n_new tokens from q.processed on; `next` = the greedy next token.  want_tensor:
// item.output_tensor receives the [n_new][vocab] logits assembled from the last group's vocabulary shards.
bool InferenceEngine::MultiStep(Query &q, int n_new, bool want_tensor, QueryInferenceResult &item, int &next)
{
    MultiGpu &M = *multi_;
    const int R = (int)M.plans.size(), P = M.P, V = spec_.hyper_params.vocab_size;
    const size_t shard = (size_t)V / (size_t)P;
    if (want_tensor && (size_t)n_new > M.shard_rows) {
        for (int i = (M.G - 1) * P; i < R; i++) {
            ifa_set_device(M.plans[(size_t)i].device);
            if (M.shard_dev[(size_t)i]) { ifa_free(M.shard_dev[(size_t)i]); M.shard_dev[(size_t)i] = nullptr; }
            if (ifa_malloc(&M.shard_dev[(size_t)i], (size_t)n_new * shard * 2) != IFA_OK) { EngineSetError("logits buffer: %s", ifa_last_error()); return false; }
        }
        M.shard_rows = (size_t)n_new;
    }
    std::vector<int> nexts((size_t)R, -1);
    std::vector<std::vector<uint16_t>> host((size_t)R);
    const int *toks = q.tokens.data() + q.processed;
    const int start = q.processed, slot = q.kv_slot;
    const bool ok = M.Run([&](int i) -> int {
        ifa_model *mm = M.plans[(size_t)i].model;
        int rc = ifa_model_select_kv(mm, slot);
        if (rc) return rc;
        void *lg = want_tensor ? M.shard_dev[(size_t)i] : nullptr;
        if (n_new == 1 && !lg) rc = ifa_model_tp_decode(mm, &M.topo[(size_t)i], toks[0], start, 1, &nexts[(size_t)i], nullptr);
        else rc = ifa_model_tp_prefill(mm, &M.topo[(size_t)i], toks, n_new, start, lg, &nexts[(size_t)i]);
        if (rc) return rc;
        if (lg) {
            host[(size_t)i].resize((size_t)n_new * shard);
            rc = ifa_memcpy_d2h(host[(size_t)i].data(), lg, (size_t)n_new * shard * 2, ifa_model_stream(mm));
            if (!rc) rc = ifa_stream_sync(ifa_model_stream(mm));
        }
        return rc;
    }, n_new == 1 ? "decode step" : "prompt step");
    if (!ok) return false;
    next = nexts[(size_t)(R - 1)];
    for (int i = 0; i < R; i++)
        if (nexts[(size_t)i] != next) { EngineSetError("ranks disagree on the next token (%d vs %d)", nexts[(size_t)i], next); return false; }
    if (want_tensor) {
        item.output_rows = n_new; item.output_cols = V;
        item.output_tensor.resize((size_t)n_new * V);
        for (int r = 0; r < P; r++) {
            const std::vector<uint16_t> &h = host[(size_t)((M.G - 1) * P + r)];
            for (int row = 0; row < n_new; row++)
                memcpy(&item.output_tensor[(size_t)row * V + (size_t)r * shard], &h[(size_t)row * shard], shard * 2);
        }
    }
    return true;
}

// One batched decode step of n queries over the (single) tensor-parallel device group: every rank's thread calls
// ifa_model_tp_decode_batch with the same rows; `all` (if wanted) receives the [n][vocab] logits assembled from the ranks'
// vocabulary shards.
bool InferenceEngine::MultiBatchStep(const std::vector<int> &toks, const std::vector<int> &pos, const std::vector<int> &slots,
                                     std::vector<int> &next, bool want_tensor, std::vector<uint16_t> &all)
{
    MultiGpu &M = *multi_;
    const int R = (int)M.plans.size(), P = M.P, V = spec_.hyper_params.vocab_size, n = (int)toks.size();
    const size_t shard = (size_t)V / (size_t)P;
    if (want_tensor && (size_t)n > M.shard_rows) {
        for (int i = (M.G - 1) * P; i < R; i++) {
            ifa_set_device(M.plans[(size_t)i].device);
            if (M.shard_dev[(size_t)i]) { ifa_free(M.shard_dev[(size_t)i]); M.shard_dev[(size_t)i] = nullptr; }
            if (ifa_malloc(&M.shard_dev[(size_t)i], (size_t)n * shard * 2) != IFA_OK) { EngineSetError("logits buffer: %s", ifa_last_error()); return false; }
        }
        M.shard_rows = (size_t)n;
    }
    std::vector<std::vector<int>> nexts((size_t)R, std::vector<int>((size_t)n, -1));
    std::vector<std::vector<uint16_t>> host((size_t)R);
    const bool ok = M.Run([&](int i) -> int {
        ifa_model *mm = M.plans[(size_t)i].model;
        void *lg = want_tensor ? M.shard_dev[(size_t)i] : nullptr;
        int rc = ifa_model_tp_decode_batch(mm, &M.topo[(size_t)i], n, toks.data(), pos.data(), slots.data(), nexts[(size_t)i].data(), lg);
        if (rc) return rc;
        if (lg) {
            host[(size_t)i].resize((size_t)n * shard);
            rc = ifa_memcpy_d2h(host[(size_t)i].data(), lg, (size_t)n * shard * 2, ifa_model_stream(mm));
            if (!rc) rc = ifa_stream_sync(ifa_model_stream(mm));
        }
        return rc;
    }, "batched decode step");
    if (!ok) return false;
    next = nexts[(size_t)(R - 1)];
    for (int i = 0; i < R; i++)
        if (nexts[(size_t)i] != next) { EngineSetError("ranks disagree on the next tokens of a batched step"); return false; }
    if (want_tensor) {
        all.resize((size_t)n * V);
        for (int r = 0; r < P; r++) {
            const std::vector<uint16_t> &h = host[(size_t)((M.G - 1) * P + r)];
            for (int row = 0; row < n; row++) memcpy(&all[(size_t)row * V + (size_t)r * shard], &h[(size_t)row * shard], shard * 2);
        }
    }
    return true;
}

int InferenceEngine::AddQuery(const std::vector<int> &tokens, const QueryOptions &query_options)
{
    if (!model_) { EngineSetError("The engine is not initialized"); return -1; }
    if (tokens.empty()) { EngineSetError("Empty query"); return -1; }
    const int max_ctx = spec_.max_context_len > 0 ? spec_.max_context_len : ModelSpec::DEFAULT_MAX_CONTEXT_LEN;
    if ((int)tokens.size() >= max_ctx) { EngineSetError("The query has %zu tokens; max_context_len is %d", tokens.size(), max_ctx); return -1; }
    for (int t : tokens)
        if (t < 0 || t >= spec_.hyper_params.vocab_size) { EngineSetError("Token id %d is out of range", t); return -1; }
    SamplingStrategyId strategy = (SamplingStrategyId)query_options.strategy_id;
    if (query_options.strategy_id < 0 || query_options.strategy_id > (int)SamplingStrategyId::Mirostat) { EngineSetError("Invalid strategy id %d", query_options.strategy_id); return -1; }
    if (strategy == SamplingStrategyId::Auto) strategy = default_strategy_;
    if (!IsSupportedStrategy(strategy)) { EngineSetError("Decoding strategy %d is not supported", query_options.strategy_id); return -1; }
    if ((int)queries_.size() >= std::min(config_.max_concurrent_queries, kv_slots_)) return 0;      // busy
    Query q; q.id = next_query_id_++; q.tokens = tokens; q.options = query_options;
    q.strategy = strategy; q.sampling = default_sampling_;
    if (query_options.random_seed != 0) q.rng.SetSeed((uint64_t)(int64_t)query_options.random_seed);    // SamplingStrategy::BeginQuery
    std::vector<bool> used((size_t)kv_slots_, false);
    for (const auto &kv : queries_) used[(size_t)kv.second.kv_slot] = true;
    while (q.kv_slot < kv_slots_ && used[(size_t)q.kv_slot]) q.kv_slot++;
    queries_[q.id] = q;
    return q.id;
}

int InferenceEngine::QueryCount() const { return (int)queries_.size(); }
int InferenceEngine::PartitionRanks() const { return multi_ ? (int)multi_->plans.size() : 1; }
ifa_model *InferenceEngine::worker(int rank)
{
    if (!multi_) return rank == 0 ? model_ : nullptr;
    return rank >= 0 && rank < (int)multi_->plans.size() ? multi_->plans[(size_t)rank].model : nullptr;
}
bool InferenceEngine::WorkerPlanOf(int rank, int out6[6]) const
{
    if (!multi_) {
        if (rank != 0) return false;
        out6[0] = 0; out6[1] = 1; out6[2] = 0; out6[3] = 1; out6[4] = 0; out6[5] = spec_.hyper_params.decoder_layers;
        return true;
    }
    if (rank < 0 || rank >= (int)multi_->plans.size()) return false;
    const WorkerPlan &w = multi_->plans[(size_t)rank];
    out6[0] = w.stage; out6[1] = w.n_stages; out6[2] = w.tp_rank; out6[3] = w.tp_size; out6[4] = w.layer0; out6[5] = w.layer1;
    return true;
}

SamplingStrategyId InferenceEngine::GetSamplingStrategyId(const std::string &str) const
{
    if (str.empty()) return default_strategy_;
    return SamplingStrategyIdFromName(str);
}

// SampleTokens (inference_engine.cc:1986-2042) for the non-greedy strategies: the logits row comes to the host
bool InferenceEngine::SampleRow(Query &q, const uint16_t *logits_row, QueryInferenceResult &item)
{
    SamplingOutput out;
    if (!ChooseTokens(out, logits_row, spec_.hyper_params.vocab_size, q.strategy, q.sampling, q.options.temperature, q.rng,
                      q.sampling_state, q.tokens) || out.selected.empty()) {
        EngineSetError("Sampling failed for query %d", q.id); return false;
    }
    item.next_tokens.clear();
    item.next_tokens.push_back(out.selected[0]);
    return true;
}

bool InferenceEngine::RemoveQuery(int query_id)
{
    return queries_.erase(query_id) != 0;
}

bool InferenceEngine::QueryEnded(int query_id) const
{
    auto it = queries_.find(query_id);
    return it == queries_.end() || it->second.ended;
}

bool InferenceEngine::Infer(InferenceResult &res)
{
    res.items.clear(); res.perf_stat.time_map.clear();
    if (!model_) { EngineSetError("The engine is not initialized"); return false; }
    const auto t0 = std::chrono::steady_clock::now();
    const int V = spec_.hyper_params.vocab_size;
    const int max_ctx = spec_.max_context_len > 0 ? spec_.max_context_len : ModelSpec::DEFAULT_MAX_CONTEXT_LEN;
    // dynamic batching: every query that advances by exactly one token joins ONE step -- the linear layers stream
    // the weights once for all of them (ifa_model_decode_batch); prefills and single queries take the paths below
    std::vector<Query *> batch;
    for (auto &kv : queries_) {
        Query &q = kv.second;
        // (a partition with several layer groups steps its queries one by one: a batched step is one tensor-parallel group's)
        const bool batchable = !multi_ || multi_->G == 1;
        if (batchable && !q.ended && (int)q.tokens.size() < max_ctx && q.processed > 0 && (int)q.tokens.size() - q.processed == 1) batch.push_back(&q);
    }
    if ((int)batch.size() >= std::max(2, config_.dynamic_batching_min_queries)) {
        const int n = (int)batch.size();
        std::vector<int> toks((size_t)n), pos((size_t)n), slots((size_t)n), next((size_t)n, -1);
        for (int r = 0; r < n; r++) { toks[(size_t)r] = batch[(size_t)r]->tokens.back(); pos[(size_t)r] = batch[(size_t)r]->processed; slots[(size_t)r] = batch[(size_t)r]->kv_slot; }
        void *lg = nullptr;
        bool any_sampled = false;
        for (Query *bq : batch) any_sampled = any_sampled || bq->strategy != SamplingStrategyId::Greedy || host_greedy_;
        std::vector<uint16_t> all;
        if (multi_) {
            // Query batching over a tensor-parallel device group (the reference: query batching, inference_engine.cc:1054-1124,
            // inside Infer_TensorParallelism, :1222-1296): every rank runs ONE batched step over the same queries -- merges
            // over [n][dim], one distributed argmax per row (ifa_model_tp_decode_batch) -- and hands back its vocabulary shard
            if (!MultiBatchStep(toks, pos, slots, next, config_.return_output_tensors || any_sampled, all)) return false;
        } else {
        if (config_.return_output_tensors || any_sampled) {
            if ((size_t)n > logits_rows_) {
                if (logits_dev_) ifa_free(logits_dev_);
                logits_dev_ = nullptr; logits_rows_ = 0;
                if (ifa_malloc(&logits_dev_, (size_t)n * V * 2) != IFA_OK) { EngineSetError("logits buffer: %s", ifa_last_error()); return false; }
                logits_rows_ = (size_t)n;
            }
            lg = logits_dev_;
        }
        if (ifa_model_decode_batch(model_, n, toks.data(), pos.data(), slots.data(), next.data(), lg) != IFA_OK) {
            EngineSetError("batched decode step failed: %s", ifa_last_error()); return false;
        }
        if (lg) {
            all.resize((size_t)n * V);
            if (ifa_memcpy_d2h(all.data(), lg, all.size() * 2, ifa_model_stream(model_)) != IFA_OK || ifa_stream_sync(ifa_model_stream(model_)) != IFA_OK) {
                EngineSetError("logits copy: %s", ifa_last_error()); return false;
            }
        }
        }
        for (int r = 0; r < n; r++) {
            Query &q = *batch[(size_t)r];
            QueryInferenceResult item; item.query_id = q.id; item.prefix_len = q.processed;
            if (!all.empty() && config_.return_output_tensors) { item.output_rows = 1; item.output_cols = V; item.output_tensor.assign(all.begin() + (size_t)r * V, all.begin() + (size_t)(r + 1) * V); }
            q.processed = (int)q.tokens.size();
            IdWeight w; w.id = next[(size_t)r]; w.weight = 1.0f;
            item.next_tokens.push_back(w);
            if ((q.strategy != SamplingStrategyId::Greedy || host_greedy_) && !SampleRow(q, all.data() + (size_t)r * V, item)) return false;
            res.items.push_back(std::move(item));
        }
    }
    for (auto &kv : queries_) {
        Query &q = kv.second;
        if (q.ended) continue;
        if ((int)q.tokens.size() >= max_ctx) { q.ended = true; continue; }
        const int n_new = (int)q.tokens.size() - q.processed;
        if (n_new <= 0) continue;                                    // nothing committed since the last step
        QueryInferenceResult item; item.query_id = q.id; item.prefix_len = q.processed;
        int next = -1;
        const bool sampled = q.strategy != SamplingStrategyId::Greedy || host_greedy_;
        const bool want_tensor = config_.return_output_tensors || sampled;
        if (multi_) {                                                // partition over several GPUs: every rank steps at once
            if (!MultiStep(q, n_new, want_tensor, item, next)) return false;
            if (sampled && !SampleRow(q, item.output_tensor.data() + (size_t)(n_new - 1) * V, item)) return false;
            if (!config_.return_output_tensors) { item.output_tensor.clear(); item.output_rows = item.output_cols = 0; }
            q.processed = (int)q.tokens.size();
            if (item.next_tokens.empty()) { IdWeight w; w.id = next; w.weight = 1.0f; item.next_tokens.push_back(w); }
            res.items.push_back(std::move(item));
            continue;
        }
        if (ifa_model_select_kv(model_, q.kv_slot) != IFA_OK) { EngineSetError("select_kv: %s", ifa_last_error()); return false; }
        if (n_new == 1 && !want_tensor) {                            // decode: fused graph-replayed step
            if (ifa_model_decode(model_, q.tokens.back(), q.processed, 1, &next, nullptr) != IFA_OK) {
                EngineSetError("decode step failed: %s", ifa_last_error()); return false;
            }
        } else {
            void *lg = nullptr;
            if (want_tensor) {
                if ((size_t)n_new > logits_rows_) {
                    if (logits_dev_) ifa_free(logits_dev_);
                    logits_dev_ = nullptr; logits_rows_ = 0;
                    if (ifa_malloc(&logits_dev_, (size_t)n_new * V * 2) != IFA_OK) { EngineSetError("logits buffer: %s", ifa_last_error()); return false; }
                    logits_rows_ = (size_t)n_new;
                }
                lg = logits_dev_;
            }
            if (ifa_model_forward(model_, q.tokens.data() + q.processed, n_new, q.processed, lg, &next) != IFA_OK) {
                EngineSetError("forward step failed: %s", ifa_last_error()); return false;
            }
            std::vector<uint16_t> last_row;
            if (config_.return_output_tensors) {
                item.output_rows = n_new; item.output_cols = V;
                item.output_tensor.resize((size_t)n_new * V);
                if (ifa_memcpy_d2h(item.output_tensor.data(), lg, (size_t)n_new * V * 2, ifa_model_stream(model_)) != IFA_OK
                    || ifa_stream_sync(ifa_model_stream(model_)) != IFA_OK) { EngineSetError("logits copy: %s", ifa_last_error()); return false; }
            } else if (sampled) {                                    // sampling only: the last row is all the host needs
                last_row.resize((size_t)V);
                if (ifa_memcpy_d2h(last_row.data(), (const uint16_t *)lg + (size_t)(n_new - 1) * V, (size_t)V * 2, ifa_model_stream(model_)) != IFA_OK
                    || ifa_stream_sync(ifa_model_stream(model_)) != IFA_OK) { EngineSetError("logits copy: %s", ifa_last_error()); return false; }
            }
            if (sampled) {
                const uint16_t *row = config_.return_output_tensors ? item.output_tensor.data() + (size_t)(n_new - 1) * V : last_row.data();
                if (!SampleRow(q, row, item)) return false;
            }
        }
        q.processed = (int)q.tokens.size();
        if (item.next_tokens.empty()) { IdWeight w; w.id = next; w.weight = 1.0f; item.next_tokens.push_back(w); }
        res.items.push_back(std::move(item));
    }
    if (perf_phases_) {
        int keys[256]; float ms[256]; int n = 0;
        if (ifa_model_perf_stat(model_, keys, ms, 256, &n, 1) != IFA_OK) { EngineSetError("perf_stat: %s", ifa_last_error()); return false; }
        for (int i = 0; i < std::min(n, 256); i++) res.perf_stat.time_map[keys[i]] = ms[i];
    }
    if (config_.debug.enable_perf_stat)
        res.perf_stat.time_map[0] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

bool InferenceEngine::CommitInferenceResult(const std::map<int, QueryNextToken> &query_map)
{
    bool ok = true;
    for (const auto &kv : query_map) {
        auto it = queries_.find(kv.first);
        if (it == queries_.end()) { EngineSetError("Query %d does not exist", kv.first); ok = false; continue; }
        Query &q = it->second;
        if (kv.second.id < 0 || kv.second.id >= spec_.hyper_params.vocab_size) { EngineSetError("Token id %d is out of range", kv.second.id); ok = false; continue; }
        q.tokens.push_back(kv.second.id);
        if (kv.second.is_end) q.ended = true;
    }
    return ok;
}

bool InferenceEngine::Generate(int query_id, int n_steps, std::vector<int> &new_tokens, float *gpu_ms)
{
    new_tokens.clear();
    auto it = queries_.find(query_id);
    if (!model_ || it == queries_.end()) { EngineSetError("Query %d does not exist", query_id); return false; }
    Query &q = it->second;
    if (n_steps <= 0) return true;
    if (q.ended) { EngineSetError("Query %d has ended", query_id); return false; }
    if (host_greedy_) { EngineSetError("Generate() decodes on the device, whose argmax excludes at most 3 token ids; this vocabulary has %zu (use Infer / CommitInferenceResult)", default_sampling_.excluded_ids.size()); return false; }
    if (q.strategy != SamplingStrategyId::Greedy) { EngineSetError("Generate() decodes greedily on the device; query %d uses strategy %d (use Infer / CommitInferenceResult)", query_id, (int)q.strategy); return false; }
    {   // nothing is touched unless the whole request fits (the device token ring holds 1024 steps per call)
        const int max_ctx = spec_.max_context_len > 0 ? spec_.max_context_len : ModelSpec::DEFAULT_MAX_CONTEXT_LEN;
        if ((int)q.tokens.size() + n_steps > max_ctx) { EngineSetError("Generate: %zu tokens + %d steps exceed max_context_len %d", q.tokens.size(), n_steps, max_ctx); return false; }
    }
    if (multi_) {
        MultiGpu &M = *multi_;
        const int R = (int)M.plans.size();
        const int pending = (int)q.tokens.size() - q.processed;
        if (pending <= 0) { EngineSetError("Query %d has no committed token to continue from", query_id); return false; }
        int next = -1;
        if (pending > 1 || q.processed == 0) {
            QueryInferenceResult item;
            if (!MultiStep(q, pending, false, item, next)) return false;
            q.processed = (int)q.tokens.size();
            q.tokens.push_back(next); new_tokens.push_back(next);
            n_steps--;
        }
        float ms_total = 0;
        while (n_steps > 0) {
            const int k = std::min(n_steps, 1024);
            std::vector<std::vector<int>> outs((size_t)R, std::vector<int>((size_t)k));
            std::vector<float> ms((size_t)R, 0.0f);
            const int first = q.tokens.back(), start = q.processed, slot = q.kv_slot;
            if (!M.Run([&](int i) -> int {
                    ifa_model *mm = M.plans[(size_t)i].model;
                    int rc = ifa_model_select_kv(mm, slot);
                    return rc ? rc : ifa_model_tp_decode(mm, &M.topo[(size_t)i], first, start, k, outs[(size_t)i].data(), &ms[(size_t)i]);
                }, "decode")) return false;
            for (int t : outs[(size_t)(R - 1)]) { q.tokens.push_back(t); new_tokens.push_back(t); }
            q.processed = (int)q.tokens.size() - 1;
            ms_total += *std::max_element(ms.begin(), ms.end());
            n_steps -= k;
        }
        if (gpu_ms) *gpu_ms = ms_total;
        return true;
    }
    if (ifa_model_select_kv(model_, q.kv_slot) != IFA_OK) { EngineSetError("select_kv: %s", ifa_last_error()); return false; }
    int next = -1;
    const int pending = (int)q.tokens.size() - q.processed;
    if (pending > 1 || q.processed == 0) {            // prefill whatever is pending; yields the first new token
        if (ifa_model_forward(model_, q.tokens.data() + q.processed, pending, q.processed, nullptr, &next) != IFA_OK) {
            EngineSetError("forward step failed: %s", ifa_last_error()); return false;
        }
        q.processed = (int)q.tokens.size();
        q.tokens.push_back(next); new_tokens.push_back(next);
        n_steps--;
    } else if (pending == 0) { EngineSetError("Query %d has no committed token to continue from", query_id); return false; }
    float ms_total = 0;
    while (n_steps > 0) {                              // the device token ring holds 1024 steps per call
        const int k = std::min(n_steps, 1024);
        std::vector<int> out((size_t)k);
        float ms = 0;
        if (ifa_model_decode(model_, q.tokens.back(), q.processed, k, out.data(), &ms) != IFA_OK) {
            EngineSetError("decode failed: %s", ifa_last_error()); return false;
        }
        ms_total += ms;
        for (int t : out) { q.tokens.push_back(t); new_tokens.push_back(t); }
        q.processed = (int)q.tokens.size() - 1;
        n_steps -= k;
    }
    if (gpu_ms) *gpu_ms = ms_total;
    return true;
}

} // namespace inferflow_amd
