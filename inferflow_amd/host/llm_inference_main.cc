// ifa_llm_inference -- token-id counterpart of the reference's llm_inference tool
// (src/tools/llm_inference.cc:345-457): load the .ini, add one query, then loop
// Infer -> pick the greedy token -> CommitInferenceResult, and report prefill / decode rates.
//
//   ifa_llm_inference <config.ini> [--section transformer_engine] [--tokens 1,15043,3186]
//                     [--prompt-len N --seed S] [--max-new 64] [--generate]
// --generate uses InferenceEngine::Generate (device-side token feedback) instead of the per-step loop.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "inference_engine.h"

using namespace inferflow_amd;

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <config.ini> [--section S] [--tokens a,b,c] [--prompt-len N] [--seed S] [--max-new N] [--generate]\n", argv[0]); return 2; }
    std::string ini = argv[1], section = "transformer_engine", tokens_arg;
    int prompt_len = 16, max_new = 64; unsigned seed = 42; bool generate = false;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--section") section = next();
        else if (a == "--tokens") tokens_arg = next();
        else if (a == "--prompt-len") prompt_len = atoi(next());
        else if (a == "--seed") seed = (unsigned)atoi(next());
        else if (a == "--max-new") max_new = atoi(next());
        else if (a == "--generate") generate = true;
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    InferenceConfig cfg;
    if (!InferenceEngine::LoadConfig(cfg, ini, section)) { fprintf(stderr, "LoadConfig: %s\n", EngineLastError()); return 1; }
    InferenceEngine engine;
    auto t0 = std::chrono::steady_clock::now();
    if (!engine.Init(cfg)) { fprintf(stderr, "Init: %s\n", EngineLastError()); return 1; }
    const float load_s = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
    const ModelSpec &spec = engine.model_spec();
    fprintf(stderr, "%s: model %s loaded in %.1f s (vocab %d, dim %d, layers %d, heads %d/%d)\n", engine.Version().c_str(),
            spec.sid.c_str(), load_s, spec.hyper_params.vocab_size, spec.hyper_params.embd_dims, spec.hyper_params.decoder_layers,
            spec.hyper_params.decoder_heads, spec.hyper_params.decoder_kv_heads);

    std::vector<int> tokens;
    if (!tokens_arg.empty()) {
        for (char *p = strtok(&tokens_arg[0], ","); p; p = strtok(nullptr, ",")) tokens.push_back(atoi(p));
    } else {
        unsigned s = seed;
        for (int i = 0; i < prompt_len; i++) { s = s * 1664525u + 1013904223u; tokens.push_back(3 + (int)((s >> 8) % (unsigned)(spec.hyper_params.vocab_size - 3))); }
    }
    const int qid = engine.AddQuery(tokens, QueryOptions());
    if (qid <= 0) { fprintf(stderr, "AddQuery: %d %s\n", qid, EngineLastError()); return 1; }

    std::vector<int> out;
    double prefill_ms = 0, decode_ms = 0;
    if (generate) {
        t0 = std::chrono::steady_clock::now();
        std::vector<int> first;
        if (!engine.Generate(qid, 1, first)) { fprintf(stderr, "Generate: %s\n", EngineLastError()); return 1; }
        prefill_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        out = first;
        t0 = std::chrono::steady_clock::now();
        std::vector<int> rest;
        if (max_new > 1 && !engine.Generate(qid, max_new - 1, rest)) { fprintf(stderr, "Generate: %s\n", EngineLastError()); return 1; }
        decode_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        out.insert(out.end(), rest.begin(), rest.end());
    } else {
        for (int step = 0; step < max_new; step++) {
            t0 = std::chrono::steady_clock::now();
            InferenceResult res;
            if (!engine.Infer(res) || res.items.empty()) { fprintf(stderr, "Infer: %s\n", EngineLastError()); return 1; }
            QueryNextToken nt; nt.id = res.items[0].next_tokens[0].id;
            std::map<int, QueryNextToken> commit; commit[qid] = nt;
            if (!engine.CommitInferenceResult(commit)) { fprintf(stderr, "Commit: %s\n", EngineLastError()); return 1; }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            (step == 0 ? prefill_ms : decode_ms) += ms;
            out.push_back(nt.id);
            if (engine.config().debug.is_study_mode && engine.config().debug.enable_perf_stat && (step == 0 || step + 1 == max_new)) {
                printf("perf_stat (step %d):\n", step);          // (the reference's tool writes the same lines to perf_stat.txt, llm_inference.cc:58-73)
                InferenceEngine::PrintPerfStat(stdout, res.perf_stat);
            }
        }
    }
    printf("prompt:");
    for (int t : tokens) printf(" %d", t);
    printf("\noutput:");
    for (int t : out) printf(" %d", t);
    printf("\nprefill: %zu tokens in %.2f ms (%.1f tok/s)\n", tokens.size(), prefill_ms, tokens.size() * 1e3 / prefill_ms);
    if (out.size() > 1) printf("decode: %zu tokens in %.2f ms (%.1f tok/s)\n", out.size() - 1, decode_ms, (out.size() - 1) * 1e3 / decode_ms);
    engine.RemoveQuery(qid);
    return 0;
}
