// sampling_strategy.cc -- see sampling_strategy.h
#include "sampling_strategy.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <ctime>
#include <map>

#include "half_bits.h"
#include "ifa_json.h"

namespace inferflow_amd {

JavaRandom::JavaRandom() { SetSeed((uint64_t)time(nullptr)); }

SamplingStrategyId SamplingStrategyIdFromName(const std::string &name)
{
    static const std::map<std::string, SamplingStrategyId> names = {
        {"std", SamplingStrategyId::StdSampling}, {"sample.std", SamplingStrategyId::StdSampling},
        {"greedy", SamplingStrategyId::Greedy}, {"sample.greedy", SamplingStrategyId::Greedy},
        {"top_k", SamplingStrategyId::TopK}, {"sample.top_k", SamplingStrategyId::TopK},
        {"top_p", SamplingStrategyId::TopP}, {"sample.top_p", SamplingStrategyId::TopP},
        {"fsd", SamplingStrategyId::FSD}, {"sample.fsd", SamplingStrategyId::FSD},
        {"random_fsd", SamplingStrategyId::RandomizedFSD}, {"sample.random_fsd", SamplingStrategyId::RandomizedFSD},
        {"min_p", SamplingStrategyId::MinP}, {"tfs", SamplingStrategyId::TFS}, {"typical", SamplingStrategyId::Typical},
        {"mirostat", SamplingStrategyId::Mirostat}};
    auto it = names.find(name);
    return it == names.end() ? SamplingStrategyId::Auto : it->second;
}

bool ParseDecodingStrategy(const std::string &text, SamplingStrategyId &id, StdSamplingConfig &cfg, std::string *err)
{
    id = SamplingStrategyId::Auto;
    size_t a = 0;
    while (a < text.size() && isspace((unsigned char)text[a])) a++;
    if (a == text.size()) return true;                          // empty: Auto (greedy here)
    if (text[a] != '{') {
        size_t b = text.size();
        while (b > a && isspace((unsigned char)text[b - 1])) b--;
        id = SamplingStrategyIdFromName(text.substr(a, b - a));
        if (id == SamplingStrategyId::Auto) { if (err) *err = "Invalid decoding_strategy"; return false; }
        return true;
    }
    JsonValue doc; JsonParser parser; std::string perr;
    if (!parser.Parse(text, doc, &perr)) { if (err) *err = "Invalid JSON format in the decoding strategy configuration: " + perr; return false; }
    std::string name;
    if (!doc.GetString("name", name)) { if (err) *err = "The \"name\" field is missing in the decoding strategy configuration"; return false; }
    id = SamplingStrategyIdFromName(name);
    if (id == SamplingStrategyId::Auto) { if (err) *err = "Invalid decoding_strategy"; return false; }
    doc.GetNumber("min_k", cfg.min_k); doc.GetNumber("max_k", cfg.max_k); doc.GetNumber("top_p", cfg.top_p);
    doc.GetNumber("pool_size", cfg.pool_size); doc.GetNumber("eos_bypassing_max", cfg.eos_bypassing_max);
    doc.GetNumber("min_p", cfg.min_p); doc.GetNumber("z", cfg.tfs_z); doc.GetNumber("p", cfg.typical_p);      // (extension: the reference
    doc.GetNumber("eta", cfg.mirostat_eta); doc.GetNumber("tau", cfg.mirostat_tau);                          //  fixes these at their defaults)
    return true;
}

void SortedTopK(const uint16_t *logits, int n, int k, std::vector<IdWeight> &pool, const std::vector<int> &excluded)
{
    pool.clear();
    if (n <= 0 || k <= 0) return;
    // "a ranks before b": higher logit, or the same logit and the lower id
    auto before = [](const IdWeight &a, const IdWeight &b) { return a.weight > b.weight || (a.weight == b.weight && a.id < b.id); };
    // bounded heap whose top is the worst kept item
    std::vector<IdWeight> heap;
    heap.reserve((size_t)k + 1);
    for (int i = 0; i < n; i++) {
        if (!excluded.empty() && std::find(excluded.begin(), excluded.end(), i) != excluded.end()) continue;
        IdWeight it; it.id = i; it.weight = HalfBitsToFloat(logits[i]);
        if (it.weight != it.weight) continue;                   // NaN never enters an ordered set
        if ((int)heap.size() < k) { heap.push_back(it); std::push_heap(heap.begin(), heap.end(), before); }
        else if (before(it, heap.front())) {
            std::pop_heap(heap.begin(), heap.end(), before);
            heap.back() = it;
            std::push_heap(heap.begin(), heap.end(), before);
        }
    }
    std::sort(heap.begin(), heap.end(), before);
    pool.swap(heap);
}

void SoftMaxPool(std::vector<IdWeight> &items, float temperature)
{
    if (items.empty()) return;
    // std::sort like the reference (sampling_strategy.cc:113): NOT stable -- the order of EQUAL logits in the pool is what the
    // toolchain's std::sort makes of the (weight desc, lower id first) list the top-k queue hands over
    std::sort(items.begin(), items.end(), [](const IdWeight &a, const IdWeight &b) { return a.weight > b.weight; });
    if (temperature < 0.001f) temperature = 0.001f;
    float max_value = items[0].weight;
    for (const IdWeight &it : items) max_value = std::max(max_value, it.weight);
    float sum = 0;
    for (IdWeight &it : items) { it.weight = (float)exp((it.weight - max_value) / temperature); sum += it.weight; }
    if (sum < 0.00001f) sum = 0.00001f;
    for (IdWeight &it : items) it.weight /= sum;
}

IdWeight DrawOne(JavaRandom &rng, const std::vector<IdWeight> &pool)
{
    IdWeight none;
    if (pool.empty()) return none;
    std::vector<double> base(pool.size());
    double upper = 0;
    for (size_t i = 0; i < pool.size(); i++) { base[i] = upper; upper += std::max(0.0, (double)pool[i].weight); }
    const double r = rng.NextDouble(0, upper);
    // the reference's interval search, including its tie behaviour at the interval ends
    uint32_t begin = 0, end = (uint32_t)pool.size() - 1, mid = begin;
    while (begin < end) {
        mid = (end + begin) / 2;
        if (r < base[mid]) end = mid;
        else if (r > base[mid + 1]) { begin = mid + 1; mid = begin; }
        else break;
    }
    return pool[mid];
}

// the pool cut of each strategy on the sorted, softmaxed pool (first entry = most probable)
static void CutMinP(const std::vector<IdWeight> &pool, float min_p, std::vector<IdWeight> &out)
{   // MinPSamplingStrategy::ChooseTokens, sampling_strategy.cc:716-724
    const float scale = pool[0].weight;
    out.assign(1, pool[0]);
    for (size_t i = 1; i < pool.size(); i++) {
        if (pool[i].weight < min_p * scale) break;
        out.push_back(pool[i]);
    }
}

static void CutTailFree(const std::vector<IdWeight> &pool, float z, std::vector<IdWeight> &out)
{   // TFSSamplingStrategy::ChooseTokens, sampling_strategy.cc:807-838 (the cumulative sum starts at the SECOND |second difference|)
    out.assign(1, pool[0]);
    if (pool.size() < 3) return;            // the reference sizes its difference vectors pool - 1 / pool - 2: undefined below 3
    std::vector<float> d1(pool.size() - 1), d2(pool.size() - 2);
    for (size_t i = 0; i < d1.size(); i++) d1[i] = pool[i].weight - pool[i + 1].weight;
    for (size_t i = 0; i < d2.size(); i++) d2[i] = std::abs(d1[i] - d1[i + 1]);
    float sum = 0.0f;
    for (float v : d2) sum += v;
    if (sum > 1e-6f) { for (float &v : d2) v /= sum; }
    else { for (float &v : d2) v = 1.0f / (float)d2.size(); }
    float cum = 0.0f;
    for (size_t i = 1; i < d2.size(); i++) {
        cum += d2[i];
        if (cum > z) break;
        out.push_back(pool[i]);
    }
}

static void CutTypical(const std::vector<IdWeight> &pool, float p, std::vector<IdWeight> &out)
{   // TypicalSamplingStrategy::ChooseTokens, sampling_strategy.cc:921-951 (the first kept entry is not counted in the mass)
    float entropy = 0.0f;
    for (const IdWeight &it : pool) entropy += -it.weight * logf(it.weight);
    std::vector<float> shifted(pool.size());
    for (size_t i = 0; i < pool.size(); i++) shifted[i] = fabsf(-logf(pool[i].weight) - entropy);
    std::vector<size_t> idx(pool.size());
    for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return shifted[a] < shifted[b]; });        // (std::sort: sampling_strategy.cc:935)
    out.assign(1, pool[idx[0]]);
    float cum = 0.0f;
    for (size_t i = 1; i < idx.size(); i++) {
        cum += pool[idx[i]].weight;
        if (cum > p) break;
        out.push_back(pool[idx[i]]);
    }
}

void NGramModel::Initialize(const std::vector<int> &tokens)
{
    tokens_ = tokens;
    following_.assign((size_t)std::max(n_, 1), {});
    const int len = (int)tokens.size();
    for (int order = 1; order <= n_; order++)
        for (int i = 0; i + order <= len; i++)
            following_[(size_t)order - 1][std::vector<int>(tokens.begin() + i, tokens.begin() + i + order - 1)].push_back(tokens[(size_t)(i + order - 1)]);
}

void NGramModel::Update(int new_token)
{
    if (following_.empty()) following_.assign((size_t)std::max(n_, 1), {});   // (the reference indexes the unsized vector here)
    for (int i = 0; i < n_; i++) {
        if ((int)tokens_.size() < i) continue;
        following_[(size_t)i][std::vector<int>(tokens_.end() - i, tokens_.end())].push_back(new_token);
    }
    tokens_.push_back(new_token);
}

std::map<int, float> NGramModel::Penalize(const std::vector<int> &candidates)
{
    std::map<int, float> penalty;
    if ((int)tokens_.size() < n_ - 1 || following_.empty()) return penalty;
    for (int cand : candidates) {
        float remaining = 1, score = 0;
        for (int i = n_ - 1; i >= 0; i--) {
            const std::vector<int> key(tokens_.end() - i, tokens_.end());
            const std::vector<int> &next = following_[(size_t)i][key];
            int count = 0;
            for (int v : next) count += v == cand;
            if (count == 0) continue;                      // (remaining is NOT reduced for a context that never saw the candidate)
            const int total = (int)next.size();
            if (i == 0) score += remaining * ((float)count / (float)total);
            else score += remaining * beta_ * ((float)count / (float)(total + 1));
            remaining = remaining - remaining * beta_;
        }
        penalty[cand] = score;                             // no stop-word list here (sw_coeff applies to none)
    }
    return penalty;
}

// get_sorted_topk (sampling_strategy.cc:17-27): the k best by (weight, lower id first)
static std::vector<IdWeight> SortedTopKOfPool(const std::vector<IdWeight> &pool, int k)
{
    std::vector<IdWeight> v(pool);
    std::stable_sort(v.begin(), v.end(), [](const IdWeight &a, const IdWeight &b) { return a.weight > b.weight || (a.weight == b.weight && a.id < b.id); });
    if ((int)v.size() > k) v.resize((size_t)std::max(k, 0));
    return v;
}

static void FsdPick(SamplingOutput &out, std::vector<IdWeight> pool, const SamplingConfig &cfg, float temperature, SamplingState &state,
                    const std::vector<int> &text)
{   // FsdSamplingStrategy::ChooseTokens, sampling_strategy.cc:476-503
    SoftMaxPool(pool, temperature);
    pool = SortedTopKOfPool(pool, cfg.fsd_k);
    std::vector<int> ids;
    for (const IdWeight &it : pool) ids.push_back(it.id);
    if (!state.fsd_started) { state.ngram = NGramModel(cfg.fsd_n, cfg.fsd_beta); state.ngram.Initialize(text); state.fsd_started = true; }
    const std::map<int, float> penalty = state.ngram.Penalize(ids);
    for (IdWeight &it : pool) {
        auto pn = penalty.find(it.id);
        if (pn != penalty.end()) it.weight = (1 - cfg.fsd_alpha) * it.weight - cfg.fsd_alpha * pn->second;
    }
    out.token_pool = SortedTopKOfPool(pool, (int)pool.size());
    out.selected.push_back(out.token_pool[0]);
}

bool ChooseTokens(SamplingOutput &out, const uint16_t *logits, int vocab, SamplingStrategyId strategy,
                  const SamplingConfig &cfg, float temperature, JavaRandom &rng, SamplingState &state,
                  const std::vector<int> &text, int eos_id)
{
    out = SamplingOutput();
    if (!logits || vocab <= 0 || !IsSupportedStrategy(strategy)) return false;
    int max_queue_len = 1;
    float top_p = 1.0f;
    if (strategy != SamplingStrategyId::Greedy) max_queue_len = std::min(cfg.pool_size, vocab);
    if (strategy == SamplingStrategyId::StdSampling || strategy == SamplingStrategyId::TopP) top_p = cfg.top_p;
    std::vector<IdWeight> pool;
    SortedTopK(logits, vocab, max_queue_len, pool, cfg.excluded_ids);
    if (pool.empty()) return true;
    const std::vector<IdWeight> raw = pool;                     // logits of the pool (Mirostat re-normalises a prefix of them)
    bool drawn = false;
    if (strategy == SamplingStrategyId::FSD) { FsdPick(out, pool, cfg, temperature, state, text); drawn = true; }
    else if (strategy == SamplingStrategyId::RandomizedFSD) {    // :587-626: after 10 tokens always FSD, before that a coin per token
        if (state.fsd_new_tokens >= 10 || rng.NextFloat(0.0f, 1.0f) >= 0.5f) { FsdPick(out, pool, cfg, temperature, state, text); drawn = true; }
        else top_p = cfg.rfsd_top_p;
    }
    if (!drawn) SoftMaxPool(pool, temperature);
    if (drawn) {
    } else if (IsStdFamily(strategy) || strategy == SamplingStrategyId::RandomizedFSD) {
        const int top_k = std::min((int)pool.size(), cfg.max_k);
        float cumulative = 0;
        for (const IdWeight &it : pool) {                        // topp_topk_filter_on_sorted, sampling_strategy.cc:29-43
            cumulative += it.weight;
            out.token_pool.push_back(it);
            if (cumulative >= top_p || (int)out.token_pool.size() >= top_k) break;
        }
    } else if (strategy == SamplingStrategyId::MinP) CutMinP(pool, cfg.min_p, out.token_pool);
    else if (strategy == SamplingStrategyId::TFS) CutTailFree(pool, cfg.tfs_z, out.token_pool);
    else if (strategy == SamplingStrategyId::Typical) CutTypical(pool, cfg.typical_p, out.token_pool);
    else {                                                       // Mirostat, sampling_strategy.cc:1036-1056
        const float mu = state.mirostat_mu == state.mirostat_mu ? state.mirostat_mu : 2.0f * cfg.mirostat_tau;
        size_t n = 0;
        while (n < pool.size() && !(-log2f(pool[n].weight) > mu)) n++;
        if (n == 0) n = 1;
        // SoftMaxPool sorted `pool`; the same order applied to the raw logits (the same std::sort on the same sequence)
        std::vector<IdWeight> prefix(raw);
        std::sort(prefix.begin(), prefix.end(), [](const IdWeight &a, const IdWeight &b) { return a.weight > b.weight; });
        prefix.resize(n);
        SoftMaxPool(prefix, temperature);
        out.token_pool = prefix;
    }
    if (out.token_pool.empty()) return true;
    if (!drawn) out.selected.push_back(DrawOne(rng, out.token_pool));
    if (eos_id >= 0) {
        if (out.token_pool[0].id == eos_id && out.selected[0].id != eos_id) out.flag = 1;
        if (cfg.eos_bypassing_max > 0 && out.selected[0].id == eos_id && out.token_pool.size() > 1
            && state.eos_bypassing_count < cfg.eos_bypassing_max) {
            out.flag = 2;
            for (const IdWeight &it : out.token_pool)
                if (it.id != eos_id) { out.selected[0] = it; state.eos_bypassing_count++; break; }
        }
    }
    if (strategy == SamplingStrategyId::Mirostat) {             // :1091-1096
        const float mu = state.mirostat_mu == state.mirostat_mu ? state.mirostat_mu : 2.0f * cfg.mirostat_tau;
        float w = out.selected[0].weight;
        for (const IdWeight &it : out.token_pool) if (it.id == out.selected[0].id) { w = it.weight; break; }
        state.mirostat_mu = mu - cfg.mirostat_eta * (-log2f(w) - cfg.mirostat_tau);
    }
    if (strategy == SamplingStrategyId::FSD || strategy == SamplingStrategyId::RandomizedFSD) {
        if (!state.fsd_started && !state.ngram.initialized()) state.ngram = NGramModel(cfg.fsd_n, cfg.fsd_beta);
        state.ngram.Update(out.selected[0].id);                  // the model follows what was SELECTED here, like the reference
        state.fsd_new_tokens++;
    }
    return true;
}

} // namespace inferflow_amd
