// model_loader.cc -- model_spec.json + weight files -> an ifa_model worker.
//
// Formats (ModelReader, src/transformer/model_reader.cc):
//   "llama2.c"     7 x int32 header + F32 tensors grouped by kind (:3248-3430)
//   "safetensors"  HF llama-style tensor names, F32 / F16 / BF16 payloads, config.json hyper-parameters
//   "synthetic"    no files: N(0, std) weights generated on the host (benchmarks / smoke tests)
// Weights are handed over as F16 and quantised ON THE DEVICE with the reference rule
// (ifa_model_set_tensor_f16 -> Quantization::QuantizeRow_*), tensor by tensor, following
// NetworkBuilder's dtype policy: device_weight_data_type for matrices of at least
// tensor_quant_threshold elements, F16 otherwise (network_builder.cc:1550-1562); the lm_head is
// quantised only for <= 20-layer models (:839-844); norms and embeddings stay F16.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>

#include "inferflow_amd.h"
#include "inference_engine.h"
#include "ifa_ini.h"
#include "ifa_json.h"

namespace inferflow_amd {

static thread_local char g_err[1024] = "";
const char *EngineLastError() { return g_err; }
void EngineSetError(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- fp conversions (round to nearest even, like the reference's half_float / __float2half_rn)
static uint16_t F32ToF16(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                 // rounds to inf
    if (x < 0x38800000u) {                                                    // subnormal half
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23);
        uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
        const int shift = 126 - e;                                            // 14..24
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = (x - 0x38000000u) >> 13;
    const uint32_t rem = x & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(sign | r);
}
static float BF16ToF32(uint16_t b) { uint32_t x = (uint32_t)b << 16; float f; memcpy(&f, &x, 4); return f; }

// ---------------------------------------------------------------- spec json
static std::string ReadFile(const std::string &path, bool *ok)
{
    std::ifstream f(path, std::ios::binary);
    *ok = (bool)f;
    std::stringstream ss; ss << f.rdbuf();
    return ss.str();
}

bool LoadModelSpecJson(ModelSpec &spec, const std::string &path)
{
    bool ok = false;
    const std::string text = ReadFile(path, &ok);
    if (!ok) { EngineSetError("Cannot open the model specification file %s", path.c_str()); return false; }
    JsonValue root; JsonParser jp; std::string err;
    if (!jp.Parse(text, root, &err) || root.type != JsonValue::Object) {
        EngineSetError("Invalid JSON in %s: %s", path.c_str(), err.c_str()); return false;
    }
    root.GetString("config_file", spec.config_file);
    if (const JsonValue *files = root.Get("model_files"))
        for (const JsonValue &f : files->arr) if (f.type == JsonValue::String) spec.model_files.push_back(f.str);
    root.GetString("model_file_format", spec.model_file_format);
    spec.model_file_format = IniConfig::Lower(spec.model_file_format);
    ModelHyperParams &hp = spec.hyper_params;
    root.GetNumber("vocab_size", hp.vocab_size);
    root.GetNumber("output_vocab_size", hp.output_vocab_size);
    root.GetNumber("qkv_format", spec.qkv_format);
    // token-id engine: the vocabulary's unk id as a number (the reference resolves "unk_token" through its tokenizer,
    // model_reader.cc:156-170, :714; StdVocabulary's default is 0); -1: none.  Plus up to two Invalid-type token ids.
    root.GetNumber("unk_token_id", spec.unk_token_id);
    if (const JsonValue *inv = root.Get("invalid_token_ids"))
        for (const JsonValue &v : inv->arr) if (v.type == JsonValue::Number) spec.invalid_token_ids.push_back((int)v.num);
    const JsonValue *ns = root.Get("network_structure");
    if (!ns || ns->type != JsonValue::Object) { EngineSetError("%s: network_structure is missing", path.c_str()); return false; }
    ns->GetString("type", spec.network_structure);
    std::string s;
    if (ns->GetString("normalization_function", s)) {
        s = IniConfig::Lower(s);
        if (s == "rms") spec.norm_alg = TensorNormAlg::RMS;
        else if (s == "std" || s == "layer_norm" || s == "standard") spec.norm_alg = TensorNormAlg::STD;
        else { EngineSetError("Invalid normalization_function: %s", s.c_str()); return false; }
    }
    if (ns->GetString("activation_function", s)) {
        s = IniConfig::Lower(s);
        if (s == "silu") spec.activation_fn = ActivationFn::SILU;
        else if (s == "gelu") spec.activation_fn = ActivationFn::GELU;
        else if (s == "relu") spec.activation_fn = ActivationFn::RELU;
        else { EngineSetError("Invalid activation_function: %s", s.c_str()); return false; }
    }
    if (ns->GetString("position_embedding", s)) {
        s = IniConfig::Lower(s);
        if (s == "rope") spec.pos_embedding_alg = PositionEmbeddingAlg::ROPE;
        else if (s == "alibi") spec.pos_embedding_alg = PositionEmbeddingAlg::ALIBI;
        else if (s.empty() || s == "empty" || s == "none") spec.pos_embedding_alg = PositionEmbeddingAlg::EMPTY;
        else { EngineSetError("Unsupported position_embedding: %s", s.c_str()); return false; }
    }
    ns->GetNumber("rope_theta", spec.rope_theta);
    ns->GetNumber("partial_rotary_factor", spec.partial_rotary_factor);
    ns->GetNumber("qk_column_order", spec.qk_column_order);
    ns->GetNumber("qkv_format", spec.qkv_format);
    ns->GetNumber("kq_scale", spec.kq_scale);
    ns->GetNumber("attn_pre_norm_base", spec.attn_pre_norm_base);
    ns->GetNumber("ffn_pre_norm_base", spec.ffn_pre_norm_base);
    ns->GetNumber("output_norm_base", spec.output_norm_base);
    ns->GetNumber("attn_out_scale", spec.attn_out_scale);
    ns->GetNumber("ffn_out_scale", spec.ffn_out_scale);
    ns->GetNumber("out_scale", spec.out_scale);
    ns->GetBool("has_embedding_linear_norm", spec.has_embedding_linear_norm);          // model_reader.cc:374-377
    ns->GetNumber("embedding_linear_scale", spec.embedding_linear_scale);
    ns->GetBool("is_parallel_attn", spec.is_parallel_attn);
    ns->GetBool("mlp_attn_share_input", spec.mlp_attn_share_input);
    ns->GetBool("is_attn_post_as_residual", spec.is_attn_post_as_residual);            // model_reader.cc (network_structure), model.h:113
    ns->GetNumber("expert_count", hp.experts);
    spec.moe_top_k_from_spec = ns->GetNumber("moe_top_k", hp.moe_top_k);
    ns->GetBool("moe_norm_top_k_prob", hp.moe_norm_top_k_prob);
    ns->GetString("tensor_name_prefix", spec.tensor_name_prefix);
    if (const JsonValue *tm = ns->Get("tensor_name_mapping"))
        for (const auto &kv : tm->obj) if (kv.second.type == JsonValue::String) spec.tensor_name_map[kv.first] = kv.second.str;
    // "synthetic": shapes come from the spec itself
    if (const JsonValue *h = root.Get("hyper_params")) {
        h->GetNumber("vocab_size", hp.vocab_size); h->GetNumber("embd_dims", hp.embd_dims);
        h->GetNumber("hidden_dim", hp.hidden_dim); h->GetNumber("decoder_layers", hp.decoder_layers);
        h->GetNumber("decoder_heads", hp.decoder_heads); h->GetNumber("decoder_kv_heads", hp.decoder_kv_heads);
        h->GetNumber("training_context_len", hp.training_context_len);
    }
    root.GetNumber("synthetic_std", spec.synthetic_std);
    return true;
}

// --------------------------------------------------------------- uploading
// layer range [layer0, layer1) of device group g of n (NetworkBuilder::SplitGpuLayers, network_builder.cc:2094-2118:
// ceil(L / n) layers per group, the last group takes the rest)
void SplitGpuLayers(int n_layers, int n_groups, std::vector<std::pair<int, int>> &ranges)
{
    ranges.clear();
    const int per = (n_layers + n_groups - 1) / n_groups;
    for (int g = 0; g < n_groups; g++) {
        const int start = g * per;
        const int end = g + 1 == n_groups ? n_layers : std::min((g + 1) * per, n_layers);
        if (end > start) ranges.push_back(std::make_pair(start, end));
    }
}

// Which slice of a tensor a worker of a multi-GPU partition holds (the reference's BY_TENSOR rule,
// src/transformer/network_builder.cc:1594-1686, device_tensor_builder.cu:203-239): wq / wk / wv / w1 / w3 and their biases
// by contiguous ROW ranges (heads, KV heads and FFN rows are split), wo / w2 by contiguous COLUMN ranges (whole quant
// blocks), the lm_head by vocabulary rows; norms, embeddings, wo / w2 biases (added once after the merge) replicated.
// Returns false when the worker does not hold the tensor at all (other layer group, embeddings / lm_head of another stage).
bool SliceForWorker(const WorkerPlan &w, int layer, int tid, size_t rows, size_t cols, TensorSlice &sl)
{
    sl = TensorSlice{0, rows, 0, cols, layer};
    if (layer >= 0) {
        if (layer < w.layer0 || layer >= w.layer1) return false;
        sl.local_layer = layer - w.layer0;
    } else {
        if (tid == IFA_T_EMBD && !w.first_stage) return false;
        if ((tid == IFA_T_OUT_NORM || tid == IFA_T_OUT_NORM_B || tid == IFA_T_LM_HEAD) && !w.last_stage) return false;
    }
    const size_t P = (size_t)std::max(1, w.tp_size), r = (size_t)w.tp_rank;
    if (P == 1) return true;
    const bool row_split = tid == IFA_T_WQ || tid == IFA_T_WK || tid == IFA_T_WV || tid == IFA_T_W1 || tid == IFA_T_W3 || (tid == IFA_T_LM_HEAD && layer < 0);
    const bool bias_split = tid == IFA_T_WQ_B || tid == IFA_T_WK_B || tid == IFA_T_WV_B || tid == IFA_T_W1_B || tid == IFA_T_W3_B;
    const bool col_split = tid == IFA_T_WO || tid == IFA_T_W2;
    if (row_split) { sl.row0 = r * (rows / P); sl.row1 = sl.row0 + rows / P; }
    else if (col_split || bias_split) { sl.col0 = r * (cols / P); sl.col1 = sl.col0 + cols / P; }
    return true;
}

namespace {

struct Uploader {
    std::vector<WorkerPlan> *plans = nullptr;
    const ModelSpec *spec = nullptr;
    std::vector<void *> dev; std::vector<size_t> dev_bytes;
    std::vector<uint16_t> staging;
    ~Uploader() { for (size_t i = 0; i < dev.size(); i++) if (dev[i]) { ifa_set_device((*plans)[i].device); ifa_free(dev[i]); } }

    static bool IsQuant(int dt) { return dt >= 7; }
    // tid: the tensor's IFA_T_* id, for its device_weight_data_type.<tensor> override (NetworkBuilder::BuildTask_Std, network_builder.cc:1551-1555)
    int MatrixType(size_t rows, size_t cols, int tid = -1) const
    {
        int wdt = spec->device_weight_data_type;
        if (tid >= 0 && tid < 40 && spec->device_weight_data_types[tid] >= 0) wdt = spec->device_weight_data_types[tid];
        if (!IsQuant(wdt)) return wdt == IFA_F32 ? IFA_F16 : wdt;       // F16 compute either way
        const int cap = ifa_block_capacity(wdt);
        if ((long long)(rows * cols) < (long long)spec->tensor_quant_threshold || cap <= 0 || cols % (size_t)cap != 0) return IFA_F16;
        return wdt;
    }
    bool Put(int layer, int tid, int target, const uint16_t *f16, size_t rows, size_t cols, int expert = -1)
    {
        if (dev.empty()) { dev.assign(plans->size(), nullptr); dev_bytes.assign(plans->size(), 0); }
        for (size_t wi = 0; wi < plans->size(); wi++) {
            WorkerPlan &w = (*plans)[wi];
            TensorSlice sl;
            if (!SliceForWorker(w, layer, tid, rows, cols, sl)) continue;
            const size_t srows = sl.row1 - sl.row0, scols = sl.col1 - sl.col0;
            int tgt = target;
            if (IsQuant(tgt)) {
                const int cap = ifa_block_capacity(tgt);
                if (cap <= 0 || scols % (size_t)cap != 0) {
                    EngineSetError("tensor %d of layer %d: %zu columns per rank do not hold whole %d-element blocks", tid, layer, scols, cap);
                    return false;
                }
            }
            const uint16_t *src = f16 + sl.row0 * cols;
            if (scols != cols) {        // column range: repack the rows contiguously
                staging.resize(srows * scols);
                for (size_t r = 0; r < srows; r++) memcpy(&staging[r * scols], f16 + (sl.row0 + r) * cols + sl.col0, scols * 2);
                src = staging.data();
            }
            const size_t bytes = srows * scols * 2;
            if (ifa_set_device(w.device) != IFA_OK) { EngineSetError("ifa_set_device(%d): %s", w.device, ifa_last_error()); return false; }
            if (bytes > dev_bytes[wi]) {
                if (dev[wi]) ifa_free(dev[wi]);
                dev[wi] = nullptr; dev_bytes[wi] = 0;
                if (ifa_malloc(&dev[wi], bytes) != IFA_OK) { EngineSetError("device allocation of %zu bytes failed: %s", bytes, ifa_last_error()); return false; }
                dev_bytes[wi] = bytes;
            }
            if (ifa_memcpy_h2d(dev[wi], src, bytes, nullptr) != IFA_OK || ifa_stream_sync(nullptr) != IFA_OK
                || ifa_model_set_tensor_f16(w.model, sl.local_layer, tid, expert, tgt, dev[wi], srows, scols) != IFA_OK) {
                EngineSetError("uploading tensor %d of layer %d (expert %d) failed: %s", tid, layer, expert, ifa_last_error());
                return false;
            }
        }
        return true;
    }
};

int RopeOrder(const ModelSpec &spec)
{
    if (spec.pos_embedding_alg != PositionEmbeddingAlg::ROPE) return 0;
    return spec.qk_column_order == 2 ? 2 : 1;          // unary_tensor_opr.h:661-735: order 2 = (c, c + dims/2) pairs
}

bool CreateWorkers(std::vector<WorkerPlan> &plans, const ModelSpec &spec)
{
    const ModelHyperParams &hp = spec.hyper_params;
    if (hp.embd_dims <= 0 || hp.decoder_layers <= 0 || hp.decoder_heads <= 0 || hp.vocab_size <= 0 || hp.hidden_dim <= 0) {
        EngineSetError("model %s: incomplete hyper-parameters", spec.sid.c_str()); return false;
    }
    if (hp.experts > 0 && (hp.moe_top_k < 1 || hp.moe_top_k > hp.experts || hp.experts > 64)) {
        EngineSetError("model %s: %d experts with moe_top_k %d is not supported (1 <= top_k <= experts <= 64)", spec.sid.c_str(), hp.experts, hp.moe_top_k);
        return false;
    }
    const int kv_heads = hp.decoder_kv_heads > 0 ? hp.decoder_kv_heads : hp.decoder_heads;
    for (WorkerPlan &w : plans) {
        if (w.layer1 < 0) {         // layer range of the worker's device group
            std::vector<std::pair<int, int>> ranges;
            SplitGpuLayers(hp.decoder_layers, std::max(1, w.n_stages), ranges);
            if ((int)ranges.size() != std::max(1, w.n_stages)) {
                EngineSetError("model %s: %d layers cannot be spread over %d device groups", spec.sid.c_str(), hp.decoder_layers, w.n_stages);
                return false;
            }
            w.layer0 = ranges[(size_t)w.stage].first; w.layer1 = ranges[(size_t)w.stage].second;
            w.first_stage = w.stage == 0; w.last_stage = w.stage + 1 == std::max(1, w.n_stages);
        }
        const int P = std::max(1, w.tp_size);
        // the reference's own constraints (network_builder.cc:1207-1213) + whole blocks per rank
        if (hp.decoder_heads % P || kv_heads % P || hp.hidden_dim % P || hp.vocab_size % P) {
            EngineSetError("model %s: heads (%d), kv heads (%d), hidden_dim (%d) and vocab (%d) must be divisible by the device-group size %d",
                           spec.sid.c_str(), hp.decoder_heads, kv_heads, hp.hidden_dim, hp.vocab_size, P);
            return false;
        }
        ifa_model_config c; memset(&c, 0, sizeof(c));
        c.dim = hp.embd_dims; c.layers = w.layer1 - w.layer0; c.heads = hp.decoder_heads / P;
        c.kv_heads = kv_heads / P;
        c.head_dim = hp.embd_dims / hp.decoder_heads; c.ffn = hp.hidden_dim / P; c.vocab = hp.vocab_size;
        c.max_ctx = spec.max_context_len > 0 ? spec.max_context_len : ModelSpec::DEFAULT_MAX_CONTEXT_LEN;
        c.norm_kind = spec.norm_alg == TensorNormAlg::RMS ? 0 : 1;
        c.act_kind = (int)spec.activation_fn; c.is_glu = 1;
        c.rope_order = RopeOrder(spec); c.use_alibi = spec.pos_embedding_alg == PositionEmbeddingAlg::ALIBI;
        c.parallel_attn = spec.is_parallel_attn; c.share_input = spec.mlp_attn_share_input;
        c.rope_theta = spec.rope_theta; c.partial_rotary = spec.partial_rotary_factor; c.kq_scale = spec.kq_scale; c.eps = 1e-5f;
        c.kv_dtype = spec.device_kv_cache_data_type == IFA_Q8_B32T2 ? IFA_Q8_B32T2 : IFA_F16;
        c.full_quant_gemv = 1; c.tp_rank = w.tp_rank; c.tp_size = P; c.device = w.device;
        c.attn_norm_base = spec.attn_pre_norm_base; c.ffn_norm_base = spec.ffn_pre_norm_base; c.out_norm_base = spec.output_norm_base;
        c.attn_out_scale = spec.attn_out_scale; c.ffn_out_scale = spec.ffn_out_scale; c.out_scale = spec.out_scale;
        // LinearNorm of the decoder input (inference_worker.cc:447-451): scale <= 0.0001 means sqrt(dim) (tensor_opr.cu:492-494)
        c.embd_scale = !spec.has_embedding_linear_norm ? 0.0f : (spec.embedding_linear_scale <= 0.0001f ? -1.0f : spec.embedding_linear_scale);
        // sparse mixture of experts (network_builder.cc:81-84, 205-209: one FFN per expert + the router "moe.gate"): every
        // rank of a device group holds its row / column slice of EVERY expert, the router is replicated
        c.experts = hp.experts; c.moe_top_k = hp.experts > 0 ? hp.moe_top_k : 0; c.moe_norm_topk = hp.moe_norm_top_k_prob ? 1 : 0;
        if (ifa_model_create(&c, &w.model) != IFA_OK) { EngineSetError("ifa_model_create (device %d): %s", w.device, ifa_last_error()); return false; }
        (void)ifa_model_set_option(w.model, "attn_post_as_residual", spec.is_attn_post_as_residual ? 1 : 0);
    }
    return true;
}

int LmHeadType(const ModelSpec &spec, size_t rows, size_t cols)
{
    const int wdt = spec.device_weight_data_type;
    if (!Uploader::IsQuant(wdt) || spec.hyper_params.decoder_layers > 20) return IFA_F16;
    const int cap = ifa_block_capacity(wdt);
    (void)rows;
    return (cap > 0 && cols % (size_t)cap == 0) ? wdt : IFA_F16;
}

// ---------------------------------------------------------------- llama2.c
bool LoadLlama2DotC(std::vector<WorkerPlan> &plans, ModelSpec &spec)
{
    const std::string path = spec.dir + (spec.model_files.empty() ? "" : spec.model_files[0]);
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { EngineSetError("Failed to open file %s", path.c_str()); return false; }
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{fp};
    uint32_t magic = 0; int version = 0;
    if (fread(&magic, 4, 1, fp) != 1) { EngineSetError("Failed to read the magic number"); return false; }
    if (magic == 0x616b3432u) {
        if (fread(&version, 4, 1, fp) != 1 || version != 1) { EngineSetError("llama2.c version %d is not supported", version); return false; }
    } else fseek(fp, 0, SEEK_SET);
    int h[7];
    if (fread(h, 4, 7, fp) != 7) { EngineSetError("Failed to read the llama2.c header"); return false; }
    ModelHyperParams &hp = spec.hyper_params;
    hp.embd_dims = h[0]; hp.hidden_dim = h[1]; hp.decoder_layers = h[2]; hp.decoder_heads = h[3]; hp.decoder_kv_heads = h[4];
    bool shared = h[5] >= 0;
    hp.vocab_size = abs(h[5]); hp.training_context_len = h[6];
    if (version == 1) {
        uint8_t v8 = 0;
        if (fread(&v8, 1, 1, fp) != 1) { EngineSetError("Failed to read the shared-classifier flag"); return false; }
        shared = v8 != 0;
        fseek(fp, 256, SEEK_SET);      // the v1 header is padded to 256 bytes
    }
    if (hp.embd_dims <= 0 || hp.decoder_heads <= 0 || hp.embd_dims % hp.decoder_heads != 0) { EngineSetError("bad llama2.c header in %s", path.c_str()); return false; }
    if (!CreateWorkers(plans, spec)) return false;
    Uploader up; up.plans = &plans; up.spec = &spec;
    const size_t D = (size_t)hp.embd_dims, F = (size_t)hp.hidden_dim, V = (size_t)hp.vocab_size, L = (size_t)hp.decoder_layers;
    const size_t HS = D / (size_t)hp.decoder_heads, KV = (size_t)hp.decoder_kv_heads * HS;
    std::vector<float> f32; std::vector<uint16_t> f16;
    auto read = [&](size_t rows, size_t cols) -> bool {
        f32.resize(rows * cols); f16.resize(rows * cols);
        if (fread(f32.data(), 4, rows * cols, fp) != rows * cols) { EngineSetError("%s is truncated", path.c_str()); return false; }
        for (size_t i = 0; i < rows * cols; i++) f16[i] = F32ToF16(f32[i]);
        return true;
    };
    std::vector<uint16_t> embd;
    if (!read(V, D)) return false;
    embd = f16;
    if (!up.Put(-1, IFA_T_EMBD, IFA_F16, f16.data(), V, D)) return false;
    struct Kind { int tid; size_t rows, cols; bool matrix; };
    const Kind kinds[] = {{IFA_T_ATTN_NORM, 1, D, false}, {IFA_T_WQ, D, D, true}, {IFA_T_WK, KV, D, true}, {IFA_T_WV, KV, D, true},
                          {IFA_T_WO, D, D, true}, {IFA_T_FFN_NORM, 1, D, false}, {IFA_T_W1, F, D, true}, {IFA_T_W2, D, F, true},
                          {IFA_T_W3, F, D, true}};
    for (const Kind &k : kinds)                 // grouped by kind: [L][rows][cols] each
        for (size_t l = 0; l < L; l++) {
            if (!read(k.rows, k.cols)) return false;
            const int target = k.matrix ? up.MatrixType(k.rows, k.cols, k.tid) : IFA_F16;
            if (!up.Put((int)l, k.tid, target, f16.data(), k.rows, k.cols)) return false;
        }
    if (!read(1, D) || !up.Put(-1, IFA_T_OUT_NORM, IFA_F16, f16.data(), 1, D)) return false;
    // model_reader.cc:3420-3421 skips training_context_len * head_size BYTES here (the legacy RoPE tables are
    // 4x that); kept as is so that unshared-classifier files load exactly what the reference loads
    fseek(fp, (long)((size_t)hp.training_context_len * HS), SEEK_CUR);
    if (!shared) { if (!read(V, D)) return false; } else f16 = embd;
    if (!up.Put(-1, IFA_T_LM_HEAD, LmHeadType(spec, V, D), f16.data(), V, D)) return false;
    return true;
}

// ------------------------------------------------------------- safetensors
struct StEntry { std::string file; std::string dtype; std::vector<size_t> shape; size_t begin = 0, end = 0, data_off = 0; };

bool IndexSafetensors(const std::string &path, std::map<std::string, StEntry> &index)
{
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { EngineSetError("Failed to open file %s", path.c_str()); return false; }
    uint64_t hlen = 0;
    bool ok = fread(&hlen, 8, 1, fp) == 1 && hlen > 0 && hlen < (1ull << 30);
    std::string header;
    if (ok) { header.resize((size_t)hlen); ok = fread(&header[0], 1, (size_t)hlen, fp) == (size_t)hlen; }
    fclose(fp);
    if (!ok) { EngineSetError("%s: bad safetensors header", path.c_str()); return false; }
    JsonValue root; JsonParser jp; std::string err;
    if (!jp.Parse(header, root, &err) || root.type != JsonValue::Object) { EngineSetError("%s: %s", path.c_str(), err.c_str()); return false; }
    for (const auto &kv : root.obj) {
        if (kv.first == "__metadata__" || kv.second.type != JsonValue::Object) continue;
        StEntry e; e.file = path; e.data_off = 8 + (size_t)hlen;
        kv.second.GetString("dtype", e.dtype);
        if (const JsonValue *sh = kv.second.Get("shape")) for (const JsonValue &d : sh->arr) e.shape.push_back((size_t)d.num);
        const JsonValue *off = kv.second.Get("data_offsets");
        if (!off || off->arr.size() != 2) { EngineSetError("%s: tensor %s has no data_offsets", path.c_str(), kv.first.c_str()); return false; }
        e.begin = (size_t)off->arr[0].num; e.end = (size_t)off->arr[1].num;
        index[kv.first] = e;
    }
    return true;
}

bool ReadStTensor(const StEntry &e, size_t rows, size_t cols, std::vector<uint16_t> &f16, const std::string &name)
{
    size_t n = 1; for (size_t d : e.shape) n *= d;
    if (n != rows * cols) { EngineSetError("tensor %s: %zu elements, expected %zu x %zu", name.c_str(), n, rows, cols); return false; }
    const size_t esz = e.dtype == "F32" ? 4 : (e.dtype == "F16" || e.dtype == "BF16") ? 2 : 0;
    if (!esz || e.end - e.begin != n * esz) { EngineSetError("tensor %s: unsupported dtype %s", name.c_str(), e.dtype.c_str()); return false; }
    FILE *fp = fopen(e.file.c_str(), "rb");
    if (!fp) { EngineSetError("Failed to open file %s", e.file.c_str()); return false; }
    std::vector<uint8_t> raw(n * esz);
    fseek(fp, (long)(e.data_off + e.begin), SEEK_SET);
    const bool ok = fread(raw.data(), 1, raw.size(), fp) == raw.size();
    fclose(fp);
    if (!ok) { EngineSetError("%s is truncated (tensor %s)", e.file.c_str(), name.c_str()); return false; }
    f16.resize(n);
    if (e.dtype == "F16") memcpy(f16.data(), raw.data(), n * 2);
    else if (e.dtype == "BF16") { const uint16_t *p = (const uint16_t *)raw.data(); for (size_t i = 0; i < n; i++) f16[i] = F32ToF16(BF16ToF32(p[i])); }
    else { const float *p = (const float *)raw.data(); for (size_t i = 0; i < n; i++) f16[i] = F32ToF16(p[i]); }
    return true;
}

bool LoadHfConfig(ModelSpec &spec)
{
    if (spec.config_file.empty()) return true;
    bool ok = false;
    const std::string text = ReadFile(spec.dir + spec.config_file, &ok);
    if (!ok) { EngineSetError("Cannot open %s%s", spec.dir.c_str(), spec.config_file.c_str()); return false; }
    JsonValue root; JsonParser jp; std::string err;
    if (!jp.Parse(text, root, &err)) { EngineSetError("%s: %s", spec.config_file.c_str(), err.c_str()); return false; }
    ModelHyperParams &hp = spec.hyper_params;
    // the key aliases ModelReader::LoadModelSpecJson accepts (model_reader.cc:470-560)
    for (const char *k : {"d_model", "n_embed", "n_embd", "hidden_size"}) root.GetNumber(k, hp.embd_dims);
    for (const char *k : {"n_layer", "num_layers", "num_hidden_layers"}) root.GetNumber(k, hp.decoder_layers);
    for (const char *k : {"n_head", "num_attention_heads"}) root.GetNumber(k, hp.decoder_heads);
    for (const char *k : {"num_kv_heads", "num_key_value_heads"}) root.GetNumber(k, hp.decoder_kv_heads);
    for (const char *k : {"ffn_hidden_size", "intermediate_size"}) root.GetNumber(k, hp.hidden_dim);
    if (hp.vocab_size <= 0) root.GetNumber("vocab_size", hp.vocab_size);
    root.GetNumber("max_position_embeddings", hp.training_context_len);
    root.GetNumber("rope_theta", spec.rope_theta);
    if (hp.decoder_kv_heads <= 0) hp.decoder_kv_heads = hp.decoder_heads;
    // mixture of experts: the spec's expert_count / moe_top_k win, config.json fills what the spec leaves open
    if (spec.network_structure.find("moe") != std::string::npos) {
        if (hp.experts <= 0) root.GetNumber("num_local_experts", hp.experts);
        int k = 0;
        if (root.GetNumber("num_experts_per_tok", k) && k > 0 && !spec.moe_top_k_from_spec) hp.moe_top_k = k;
    }
    return true;
}

bool LoadSafetensors(std::vector<WorkerPlan> &plans, ModelSpec &spec)
{
    if (!LoadHfConfig(spec)) return false;
    std::map<std::string, StEntry> index;
    for (const std::string &f : spec.model_files) {
        if (f.size() > 5 && f.compare(f.size() - 5, 5, ".json") == 0) continue;     // *.index.json
        if (!IndexSafetensors(spec.dir + f, index)) return false;
    }
    // file name -> standard name: strip the prefix, then apply tensor_name_mapping
    std::map<std::string, StEntry> by_std;
    for (const auto &kv : index) {
        std::string name = kv.first;
        if (!spec.tensor_name_prefix.empty() && name.compare(0, spec.tensor_name_prefix.size(), spec.tensor_name_prefix) == 0)
            name = name.substr(spec.tensor_name_prefix.size());
        for (const auto &mp : spec.tensor_name_map) {
            size_t p = name.find(mp.first);
            if (p != std::string::npos) name.replace(p, mp.first.size(), mp.second);
        }
        by_std[name] = kv.second;
    }
    if (!CreateWorkers(plans, spec)) return false;
    const ModelHyperParams &hp = spec.hyper_params;
    Uploader up; up.plans = &plans; up.spec = &spec;
    const size_t D = (size_t)hp.embd_dims, F = (size_t)hp.hidden_dim, V = (size_t)hp.vocab_size;
    const size_t HS = D / (size_t)hp.decoder_heads, KV = (size_t)hp.decoder_kv_heads * HS;
    std::vector<uint16_t> f16;
    auto put = [&](const std::string &name, int layer, int tid, size_t rows, size_t cols, bool matrix, bool required, int expert = -1) -> int {
        auto it = by_std.find(name);
        if (it == by_std.end()) { if (required) EngineSetError("tensor %s is missing", name.c_str()); return required ? -1 : 0; }
        if (!ReadStTensor(it->second, rows, cols, f16, name)) return -1;
        const int target = tid == IFA_T_LM_HEAD ? LmHeadType(spec, rows, cols) : (matrix ? up.MatrixType(rows, cols, tid) : IFA_F16);
        return up.Put(layer, tid, target, f16.data(), rows, cols, expert) ? 1 : -1;
    };
    // Mixtral-style checkpoints (data/models/mixtral_8x7b_instruct_v0.1/model_spec.safetensors.json): the router
    // "block_sparse_moe.gate" [experts][dim] and per expert w1 / w3 [ffn][dim], w2 [dim][ffn]; the standard names the
    // reference maps them to ("dec.{i}.moe.gate", "dec.{i}.moe.expert.{j}.w1") are accepted as well
    const int n_experts = hp.experts;
    if (put("embed_tokens.weight", -1, IFA_T_EMBD, V, D, false, true) < 0) return false;
    const std::vector<uint16_t> embd = f16;
    for (int l = 0; l < hp.decoder_layers; l++) {
        const std::string p = "layers." + std::to_string(l) + ".";
        struct E { const char *name; int tid; size_t rows, cols; bool matrix; };
        const E es[] = {{"input_layernorm.weight", IFA_T_ATTN_NORM, 1, D, false},
                        {"self_attn.q_proj.weight", IFA_T_WQ, D, D, true}, {"self_attn.k_proj.weight", IFA_T_WK, KV, D, true},
                        {"self_attn.v_proj.weight", IFA_T_WV, KV, D, true}, {"self_attn.o_proj.weight", IFA_T_WO, D, D, true},
                        {"post_attention_layernorm.weight", IFA_T_FFN_NORM, 1, D, false},
                        {"mlp.gate_proj.weight", IFA_T_W1, F, D, true}, {"mlp.down_proj.weight", IFA_T_W2, D, F, true},
                        {"mlp.up_proj.weight", IFA_T_W3, F, D, true}};
        // fused QKV checkpoints (Falcon, Bloom, GPT-NeoX ...): one [QD + 2 KVD][D] tensor whose rows are split exactly like
        // the reference splits the fused product (Attention_CalculateCurQKV, inference_worker.cc:1503-1548):
        //   qkv_format 0: per KV group {its h query heads, k, v};  qkv_format 1: all q | all k | all v
        bool fused_qkv = false;
        for (const char *fname : {"self_attn.qkv_proj.weight", "self_attention.query_key_value.weight", "attention.query_key_value.weight"}) {
            auto it = by_std.find(p + fname);
            if (it == by_std.end()) continue;
            const size_t QDr = D, total = QDr + 2 * KV;
            if (!ReadStTensor(it->second, total, D, f16, p + fname)) return false;
            const size_t groups = (size_t)hp.decoder_kv_heads, hq = (size_t)hp.decoder_heads / groups;
            std::vector<uint16_t> q(QDr * D), k(KV * D), v(KV * D);
            if (spec.qkv_format == 0) {
                for (size_t g = 0; g < groups; g++) {
                    const uint16_t *src = f16.data() + g * (hq + 2) * HS * D;
                    memcpy(q.data() + g * hq * HS * D, src, hq * HS * D * 2);
                    memcpy(k.data() + g * HS * D, src + hq * HS * D, HS * D * 2);
                    memcpy(v.data() + g * HS * D, src + (hq + 1) * HS * D, HS * D * 2);
                }
            } else {
                memcpy(q.data(), f16.data(), QDr * D * 2);
                memcpy(k.data(), f16.data() + QDr * D, KV * D * 2);
                memcpy(v.data(), f16.data() + (QDr + KV) * D, KV * D * 2);
            }
            if (!up.Put(l, IFA_T_WQ, up.MatrixType(QDr, D, IFA_T_WQ), q.data(), QDr, D) || !up.Put(l, IFA_T_WK, up.MatrixType(KV, D, IFA_T_WK), k.data(), KV, D)
                || !up.Put(l, IFA_T_WV, up.MatrixType(KV, D, IFA_T_WV), v.data(), KV, D)) return false;
            fused_qkv = true;
            break;
        }
        bool moe_layer = false;
        if (n_experts > 0) {
            const std::string ls = std::to_string(l);
            const std::string gate_names[] = {p + "block_sparse_moe.gate.weight", "dec." + ls + ".moe.gate.weight", p + "moe.gate.weight"};
            for (const std::string &gn : gate_names) {
                if (by_std.find(gn) == by_std.end()) continue;
                if (put(gn, l, IFA_T_MOE_GATE, (size_t)n_experts, D, false, true) < 0) return false;
                moe_layer = true;
                break;
            }
            if (moe_layer) {
                for (int j = 0; j < n_experts; j++) {
                    const std::string js = std::to_string(j);
                    const std::string bases[] = {p + "block_sparse_moe.experts." + js + ".", "dec." + ls + ".moe.expert." + js + ".", p + "moe.expert." + js + "."};
                    const std::string *base = nullptr;
                    for (const std::string &b : bases) if (by_std.find(b + "w1.weight") != by_std.end()) { base = &b; break; }
                    if (!base) { EngineSetError("layer %d: the tensors of expert %d are missing", l, j); return false; }
                    if (put(*base + "w1.weight", l, IFA_T_W1, F, D, true, true, j) < 0 || put(*base + "w2.weight", l, IFA_T_W2, D, F, true, true, j) < 0
                        || put(*base + "w3.weight", l, IFA_T_W3, F, D, true, true, j) < 0) return false;
                }
            }
        }
        for (const E &e : es) {
            if (fused_qkv && (e.tid == IFA_T_WQ || e.tid == IFA_T_WK || e.tid == IFA_T_WV)) continue;
            if (moe_layer && (e.tid == IFA_T_W1 || e.tid == IFA_T_W2 || e.tid == IFA_T_W3)) continue;      // the experts are this layer's FFN
            if (put(p + e.name, l, e.tid, e.rows, e.cols, e.matrix, true) < 0) return false;
        }
        // optional: biases, and the post norms in the reference's standard names (self_attn.post_norm / feed_forward.post_norm,
        // NetworkStructure::BuildTensorNameToIdMap, network_structure.cc:131-134; reached through the spec's tensor_name_mapping)
        const E bs[] = {{"self_attn.q_proj.bias", IFA_T_WQ_B, 1, D, false}, {"self_attn.k_proj.bias", IFA_T_WK_B, 1, KV, false},
                        {"self_attn.v_proj.bias", IFA_T_WV_B, 1, KV, false},
                        {"self_attn.post_norm.weight", IFA_T_ATTN_POST_NORM, 1, D, false}, {"self_attn.post_norm.bias", IFA_T_ATTN_POST_NORM_B, 1, D, false},
                        {"feed_forward.post_norm.weight", IFA_T_FFN_POST_NORM, 1, D, false}, {"feed_forward.post_norm.bias", IFA_T_FFN_POST_NORM_B, 1, D, false}};
        for (const E &e : bs) if (put(p + e.name, l, e.tid, e.rows, e.cols, false, false) < 0) return false;
    }
    if (put("norm.weight", -1, IFA_T_OUT_NORM, 1, D, false, true) < 0) return false;
    int rc = put("lm_head.weight", -1, IFA_T_LM_HEAD, V, D, true, false);
    if (rc < 0) return false;
    if (rc == 0) {       // tied embeddings
        by_std["lm_head.weight"] = by_std["embed_tokens.weight"];
        if (put("lm_head.weight", -1, IFA_T_LM_HEAD, V, D, true, true) < 0) return false;
    }
    return true;
}

// --------------------------------------------------------------- synthetic
// counter-based generator: value i of a tensor depends only on (seed, i), so the result is
// independent of the number of host threads
inline uint64_t Mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void FillNormalF16(std::vector<uint16_t> &out, size_t n, uint64_t seed, float std_dev)
{
    out.resize(n);
    const unsigned nt = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    std::vector<std::thread> ts;
    const size_t pairs = (n + 1) / 2;
    for (unsigned t = 0; t < nt; t++)
        ts.emplace_back([&, t]() {
            for (size_t p = pairs * t / nt; p < pairs * (t + 1) / nt; p++) {
                // Box-Muller in single precision (24-bit uniforms): the values end up as F16 with std 0.02, and a 34B-parameter
                // model draws 17 G pairs -- the double-precision form was two minutes of a 16-core host per model
                const uint64_t a = Mix64(seed * 0x100000001B3ull + 2 * p), b = Mix64(seed * 0x100000001B3ull + 2 * p + 1);
                const float u1 = ((float)(a >> 40) + 1.0f) * (1.0f / 16777217.0f), u2 = (float)(b >> 40) * (1.0f / 16777216.0f);
                const float r = sqrtf(-2.0f * logf(u1)), th = 6.2831853f * u2;
                out[2 * p] = F32ToF16(r * cosf(th) * std_dev);
                if (2 * p + 1 < n) out[2 * p + 1] = F32ToF16(r * sinf(th) * std_dev);
            }
        });
    for (auto &t : ts) t.join();
}

bool LoadSynthetic(std::vector<WorkerPlan> &plans, ModelSpec &spec)
{
    ModelHyperParams &hp = spec.hyper_params;
    if (hp.decoder_kv_heads <= 0) hp.decoder_kv_heads = hp.decoder_heads;
    if (!CreateWorkers(plans, spec)) return false;
    Uploader up; up.plans = &plans; up.spec = &spec;
    const size_t D = (size_t)hp.embd_dims, F = (size_t)hp.hidden_dim, V = (size_t)hp.vocab_size;
    const size_t HS = D / (size_t)hp.decoder_heads, KV = (size_t)hp.decoder_kv_heads * HS;
    std::vector<uint16_t> w, ones(D, 0x3C00);
    FillNormalF16(w, V * D, 999, spec.synthetic_std);
    if (!up.Put(-1, IFA_T_EMBD, IFA_F16, w.data(), V, D)) return false;
    if (!up.Put(-1, IFA_T_OUT_NORM, IFA_F16, ones.data(), 1, D)) return false;
    FillNormalF16(w, V * D, 998, spec.synthetic_std);
    if (!up.Put(-1, IFA_T_LM_HEAD, LmHeadType(spec, V, D), w.data(), V, D)) return false;
    for (int l = 0; l < hp.decoder_layers; l++) {
        if (!up.Put(l, IFA_T_ATTN_NORM, IFA_F16, ones.data(), 1, D) || !up.Put(l, IFA_T_FFN_NORM, IFA_F16, ones.data(), 1, D)) return false;
        struct E { int tid; size_t rows, cols; };
        const E es[] = {{IFA_T_WQ, D, D}, {IFA_T_WK, KV, D}, {IFA_T_WV, KV, D}, {IFA_T_WO, D, D}, {IFA_T_W1, F, D}, {IFA_T_W3, F, D}, {IFA_T_W2, D, F}};
        for (const E &e : es) {
            const bool ffn = e.tid == IFA_T_W1 || e.tid == IFA_T_W2 || e.tid == IFA_T_W3;
            if (hp.experts > 0 && ffn) {      // one FFN per expert (the seeds of inferflow_amd/synth.py)
                for (int j = 0; j < hp.experts; j++) {
                    FillNormalF16(w, e.rows * e.cols, 100000 + ((uint64_t)l * 64 + (uint64_t)j) * 16 + (uint64_t)e.tid, spec.synthetic_std);
                    if (!up.Put(l, e.tid, up.MatrixType(e.rows, e.cols, e.tid), w.data(), e.rows, e.cols, j)) return false;
                }
                continue;
            }
            FillNormalF16(w, e.rows * e.cols, 1000 + (uint64_t)l * 16 + (uint64_t)e.tid, spec.synthetic_std);
            if (!up.Put(l, e.tid, up.MatrixType(e.rows, e.cols, e.tid), w.data(), e.rows, e.cols)) return false;
        }
        if (hp.experts > 0) {                  // the router: always F16
            FillNormalF16(w, (size_t)hp.experts * D, 1000 + (uint64_t)l * 16 + (uint64_t)IFA_T_MOE_GATE, spec.synthetic_std);
            if (!up.Put(l, IFA_T_MOE_GATE, IFA_F16, w.data(), (size_t)hp.experts, D)) return false;
        }
    }
    return true;
}

} // namespace

bool BuildWorkers(std::vector<WorkerPlan> &plans, ModelSpec &spec)
{
    for (WorkerPlan &w : plans) w.model = nullptr;
    bool ok = false;
    const std::string &fmt = spec.model_file_format;
    // layer ranges are only known once the checkpoint's header is read: plans with layer1 < 0 are completed by FinishPlans
    if (fmt == "llama2.c" || fmt == "llama2_c") ok = LoadLlama2DotC(plans, spec);
    else if (fmt == "safetensors") ok = LoadSafetensors(plans, spec);
    else if (fmt == "synthetic") ok = LoadSynthetic(plans, spec);
    else EngineSetError("model %s: model_file_format \"%s\" is not supported (llama2.c, safetensors, synthetic)", spec.sid.c_str(), fmt.c_str());
    for (WorkerPlan &w : plans)
        if (ok && ifa_model_finalize(w.model) != IFA_OK) { EngineSetError("ifa_model_finalize: %s", ifa_last_error()); ok = false; }
    if (!ok) for (WorkerPlan &w : plans) if (w.model) { ifa_model_destroy(w.model); w.model = nullptr; }
    return ok;
}

bool BuildWorker(ifa_model **out, ModelSpec &spec, int device)
{
    *out = nullptr;
    std::vector<WorkerPlan> plans(1);
    plans[0].device = device;
    if (!BuildWorkers(plans, spec)) return false;
    *out = plans[0].model;
    return true;
}

} // namespace inferflow_amd
