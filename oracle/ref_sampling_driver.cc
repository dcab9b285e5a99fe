// ref_sampling_driver.cc -- TEST INFRASTRUCTURE (oracle/_ref): drives the reference's own decoding strategies
// (src/transformer/sampling_strategy.cc, decoding_strategies.cc -- compiled from the sources where they lie under
// /root/reference by oracle/Makefile's `ref_sampling` target) the way InferenceEngine does:
//   DecodingStrategies::Init / Get(id)                       (inference_engine.cc:1838-1841)
//   SamplingStrategy::BeginQuery(query_id, options, config)  (inference_engine.cc:397-401)
//   SamplingStrategy::ChooseTokens(output, input, vocab, id) (inference_engine.cc:1985-2022, SampleTokens)
// on one F16 logits row, n_draws times for the same query (the strategy keeps the query's generator / n-gram / mu state),
// and dumps the selected (id, weight) of every draw plus the token pool of the last one.  Used by
// tests/golden/gen_sampling_fixtures.py to pin oracle/sampling.py and host/sampling_strategy.cc to the reference.
// Nothing in inferflow_amd/ links or loads this.
//
//   ifa_ref_sampling <in.bin> <out.bin>
// in.bin : int32 magic 0x49465331, vocab, strategy_id, seed, float32 temperature, int32 n_draws, n_text, config_len,
//          uint16 logits[vocab] (F16 bits), int32 text[n_text] (the query's prompt tokens), char config[config_len] (JSON or empty)
// out.bin: int32 n_draws, (int32 id, float32 weight) per draw, int32 pool_n, (int32 id, float32 weight) per pool entry.
// Vocabulary: `vocab` Normal tokens, no unk / eos id among them (nothing excluded, no eos bypassing).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "sslib/json.h"
#include "transformer/decoding_strategies.h"
#include "transformer/transformer_types.h"

using namespace inferflow;
using namespace inferflow::transformer;

int main(int argc, const char *argv[])
{
    if (argc < 3) { fprintf(stderr, "usage: %s <in.bin> <out.bin>\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    int32_t head[4]; float temperature; int32_t tail[3];
    if (fread(head, 4, 4, f) != 4 || fread(&temperature, 4, 1, f) != 1 || fread(tail, 4, 3, f) != 3 || head[0] != 0x49465331) {
        fprintf(stderr, "bad header\n"); return 2;
    }
    const int vocab_n = head[1], strategy = head[2], seed = head[3], n_draws = tail[0], n_text = tail[1], config_len = tail[2];
    std::vector<uint16_t> bits((size_t)vocab_n);
    std::vector<int32_t> text((size_t)n_text);
    std::string config((size_t)config_len, '\0');
    if (fread(bits.data(), 2, bits.size(), f) != bits.size() || (n_text && fread(text.data(), 4, text.size(), f) != text.size()) ||
        (config_len && fread(&config[0], 1, config.size(), f) != config.size())) { fprintf(stderr, "short input\n"); return 2; }
    fclose(f);

    StdVocabulary vocab;
    vocab.token_array.resize((size_t)vocab_n);
    for (int i = 0; i < vocab_n; i++) {
        vocab.token_array[i].id = i;
        vocab.token_array[i].str = "t" + std::to_string(i);
        vocab.token_array[i].type = (int)TokenType::Normal;
    }
    vocab.SetUnk(-1); vocab.SetBos(-1); vocab.SetEos(-1);

    DecodingStrategies strategies;
    strategies.Init();
    const SamplingStrategyId id = (SamplingStrategyId)strategy;
    SamplingStrategy *st = strategies.Get(id);
    if (!st) { fprintf(stderr, "no strategy %d\n", strategy); return 2; }
    sslib::JsonParser jparser;
    jparser.Init();
    SamplingStrategy::QueryOptions opt;
    opt.strategy_id = id; opt.random_seed = seed; opt.temperature = temperature;
    const int query_id = 1;
    st->BeginQuery(query_id, opt, config, &jparser);

    std::vector<int> prefix(text.begin(), text.end()), cur;
    SamplingInput input;
    input.query_id = query_id;
    input.prefix = &prefix;
    input.cur_tokens = &cur;
    input.candidates_fp16.resize((size_t)vocab_n);
    memcpy((void *)input.candidates_fp16.data(), bits.data(), bits.size() * 2);

    FILE *o = fopen(argv[2], "wb");
    if (!o) { fprintf(stderr, "cannot write %s\n", argv[2]); return 2; }
    int32_t n = n_draws;
    fwrite(&n, 4, 1, o);
    SamplingOutput out;
    for (int d = 0; d < n_draws; d++) {
        st->ChooseTokens(out, input, vocab, id, 1);
        int32_t tok = out.selected.empty() ? -1 : (int32_t)out.selected[0].id;
        float w = out.selected.empty() ? 0.0f : out.selected[0].weight;
        fwrite(&tok, 4, 1, o); fwrite(&w, 4, 1, o);
    }
    int32_t pn = (int32_t)out.token_pool.size();
    fwrite(&pn, 4, 1, o);
    for (const auto &it : out.token_pool) {
        int32_t tok = (int32_t)it.id; float w = it.weight;
        fwrite(&tok, 4, 1, o); fwrite(&w, 4, 1, o);
    }
    fclose(o);
    st->EndQuery(query_id);
    return 0;
}
