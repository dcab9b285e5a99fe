// sampling_strategy.h -- the reference's standard decoding strategies on the host:
// Greedy, StdSampling ("sample.std"), TopK and TopP of StdSamplingStrategy
// (src/transformer/sampling_strategy.cc:235-431): top-`pool_size` of the logits in TopKQueue order,
// SoftMax with temperature over that pool, top_p / max_k cut, one draw with the engine's generator.
// The generator is sslib's Random (3rd_party/sslib/random.h:15-121) = the published java.util.Random
// LCG, so a seeded query draws the same tokens as the reference given the same logits.
// Also MinP (:696-760), TFS (:787-876), Typical (:901-990) and Mirostat (:1015-1098), which cut the same
// softmaxed pool by their own rule before the draw, and FSD / RandomizedFSD (:457-541, :569-667): the
// top-k probabilities are discounted by an n-gram model of the query's own text (NGram,
// sampling_strategy.h:125-236) and the best one is taken (RandomizedFSD alternates with top-p draws).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace inferflow_amd {

struct IdWeight { int id = 0; float weight = 0; };

enum class SamplingStrategyId {     // DecodingStrategyId, sampling_strategy.h:55-68
    Auto = 0, StdSampling, Greedy, TopK, TopP, FSD, RandomizedFSD, MinP, TFS, Typical, Mirostat
};

// "sample.std", "greedy", "sample.top_p", ... (decoding_strategies.cc:96-111); Auto when unknown
SamplingStrategyId SamplingStrategyIdFromName(const std::string &name);
inline bool IsStdFamily(SamplingStrategyId id)
{
    return id == SamplingStrategyId::StdSampling || id == SamplingStrategyId::Greedy || id == SamplingStrategyId::TopK
        || id == SamplingStrategyId::TopP;
}
inline bool IsSupportedStrategy(SamplingStrategyId id) { return (int)id >= (int)SamplingStrategyId::StdSampling && (int)id <= (int)SamplingStrategyId::Mirostat; }

class JavaRandom {                  // sslib::Random: seed' = (seed * 0x5DEECE66D + 0xB) mod 2^48
public:
    JavaRandom();                                   // seeded from the clock, like the reference's default
    explicit JavaRandom(uint64_t seed) { SetSeed(seed); }
    void SetSeed(uint64_t seed) { seed_ = (seed ^ MULTIPLIER) & MASK; }
    int32_t Next(int bits)
    {
        seed_ = (seed_ * MULTIPLIER + ADDEND) & MASK;
        return (int32_t)(seed_ >> (48 - bits));
    }
    double NextDouble() { return (double)(((int64_t)Next(26) << 27) + Next(27)) / (double)(1LL << 53); }
    double NextDouble(double from, double to) { return from + NextDouble() * (to - from); }
    float NextFloat() { return Next(24) / ((float)(1 << 24)); }
    float NextFloat(float from, float to) { return from + NextFloat() * (to - from); }

private:
    static constexpr uint64_t MULTIPLIER = 0x5DEECE66DULL, ADDEND = 0xBULL, MASK = (1ULL << 48) - 1;
    uint64_t seed_ = 0;
};

struct SamplingConfig {             // the Config structs of sampling_strategy.h (:242-249, :379-384, :419-424, :459-464, :499-505)
    int min_k = 1, max_k = 8;       // Std family
    float top_p = 0.9f;
    int pool_size = 50;             // every strategy: size of the softmaxed candidate pool
    int eos_bypassing_max = 0;
    float min_p = 0.05f;            // MinP: keep p >= min_p * p_max
    float tfs_z = 0.95f;            // TFS: mass of the normalised |second differences| to keep
    float typical_p = 0.95f;        // Typical: mass, in order of |-log p - entropy|
    float mirostat_eta = 0.1f, mirostat_tau = 5.0f;   // Mirostat v2: mu starts at 2 tau, mu -= eta (surprise - tau)
    int fsd_k = 6, fsd_n = 3;       // FSD: candidates kept, n-gram order
    float fsd_alpha = 0.5f, fsd_beta = 0.9f;          // weight' = (1 - alpha) p - alpha penalty; back-off factor
    float rfsd_top_p = 0.93f;       // RandomizedFSD's sampling branch
    std::vector<int> excluded_ids;  // ids GetSortedTopK never offers: the unk id and Invalid-type tokens (:281-297)
};
typedef SamplingConfig StdSamplingConfig;

// NGram (sampling_strategy.h:125-236): counts of what followed every (n-1)-, ..., 0-token context in the text so far
class NGramModel {
public:
    NGramModel() {}
    NGramModel(int n, float beta) : n_(n), beta_(beta) {}
    void Initialize(const std::vector<int> &tokens);
    void Update(int new_token);
    // candidate -> back-off mixture of its relative frequencies after the current contexts (empty before n - 1 tokens)
    std::map<int, float> Penalize(const std::vector<int> &candidates);
    bool initialized() const { return !following_.empty(); }

private:
    int n_ = 3;
    float beta_ = 0.9f;
    std::vector<int> tokens_;
    std::vector<std::map<std::vector<int>, std::vector<int>>> following_;      // [context length][context] -> next tokens
};

struct SamplingState {              // what the reference keeps in *QueryData besides the generator
    int eos_bypassing_count = 0;
    float mirostat_mu = __builtin_nanf("");     // unset -> 2 tau at the first draw
    NGramModel ngram;               // FSD / RandomizedFSD
    bool fsd_started = false;
    int fsd_new_tokens = 0;
};

struct SamplingOutput {
    std::vector<IdWeight> token_pool;   // probabilities after the top_p / max_k cut, descending
    std::vector<IdWeight> selected;     // the drawn token (weight = its probability in the pool)
    int flag = 0;                       // 1: the top token is EOS but another one was drawn; 2: EOS bypassed
};

// model.<name>.decoding_strategy: a name, or {"name": "sample.top_p", "top_p": 0.9, "max_k": 8, ...}
bool ParseDecodingStrategy(const std::string &text, SamplingStrategyId &id, StdSamplingConfig &cfg, std::string *err = nullptr);

// the k best (logit, id) pairs, best first; equal logits: lower id first (TopKQueue::LessWeight, top_k_queue.h:13-22)
// excluded: ids never offered to the queue -- the vocabulary's unk id and Invalid-type tokens (:281-297)
void SortedTopK(const uint16_t *logits_f16, int n, int k, std::vector<IdWeight> &pool,
                const std::vector<int> &excluded = std::vector<int>());
// SamplingStrategy::SoftMax (sampling_strategy.cc:107-147): temperature floor 0.001, sum floor 1e-5
void SoftMaxPool(std::vector<IdWeight> &items, float temperature);
// Random::RandomSampling(output, input, 1) (random.cc:75-146): one draw proportional to the weights
IdWeight DrawOne(JavaRandom &rng, const std::vector<IdWeight> &pool);

// StdSamplingStrategy::ChooseTokens (sampling_strategy.cc:359-431).  eos_id < 0: the model has no EOS notion here
// (token-id queries), the EOS flags / bypassing are skipped.  *eos_bypassing_count is the query's running count.
// state: the query's running state (Mirostat's mu, the FSD n-gram model, the EOS bypass count); text: the query's tokens so
// far (prompt + committed), read when FSD builds its n-gram model at the first call.
bool ChooseTokens(SamplingOutput &out, const uint16_t *logits_f16, int vocab, SamplingStrategyId strategy,
                  const SamplingConfig &cfg, float temperature, JavaRandom &rng, SamplingState &state,
                  const std::vector<int> &text = std::vector<int>(), int eos_id = -1);

} // namespace inferflow_amd
