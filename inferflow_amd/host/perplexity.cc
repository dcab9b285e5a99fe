// perplexity.cc -- see perplexity.h
#include "perplexity.h"

#include "half_bits.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

namespace inferflow_amd {

double TokenNll(const uint16_t *row, int vocab, int token_id)
{
    float max_logit = HalfBitsToFloat(row[0]);
    for (int i = 1; i < vocab; i++) max_logit = std::max(max_logit, HalfBitsToFloat(row[i]));
    double sum_exp = 0;
    for (int i = 0; i < vocab; i++) sum_exp += expf(HalfBitsToFloat(row[i]) - max_logit);
    return -((double)(HalfBitsToFloat(row[token_id]) - max_logit) - log(sum_exp));
}

bool ComputePerplexity(InferenceEngine &engine, const std::vector<int> &tokens, int max_length, int stride,
                       PerplexityResult &out, int host_threads)
{
    out = PerplexityResult();
    if (max_length < 2 || stride < 1) { EngineSetError("perplexity: max_length must be >= 2 and stride >= 1"); return false; }
    if (engine.QueryCount() != 0) { EngineSetError("perplexity: the engine has active queries"); return false; }
    const ModelSpec &spec = engine.model_spec();
    const int V = spec.hyper_params.vocab_size;
    const int max_ctx = spec.max_context_len > 0 ? spec.max_context_len : ModelSpec::DEFAULT_MAX_CONTEXT_LEN;
    if (max_length >= max_ctx) { EngineSetError("perplexity: max_length %d does not fit max_context_len %d", max_length, max_ctx); return false; }
    for (int t : tokens)
        if (t < 0 || t >= V) { EngineSetError("perplexity: token id %d is out of range", t); return false; }
    const int n_tokens = (int)tokens.size();
    host_threads = std::max(1, std::min(host_threads, 64));
    for (int start = 0; start < n_tokens; start += stride) {
        const int end = std::min(start + max_length, n_tokens);
        const std::vector<int> window(tokens.begin() + start, tokens.begin() + end);
        const int qid = engine.AddQuery(window, QueryOptions());
        if (qid <= 0) { if (qid == 0) EngineSetError("perplexity: engine busy"); return false; }
        InferenceResult res;
        if (!engine.Infer(res) || res.items.size() != 1) { engine.RemoveQuery(qid); if (res.items.size() != 1) EngineSetError("perplexity: no result for the window at %d", start); return false; }
        const QueryInferenceResult &item = res.items[0];
        if (item.output_rows != (int)window.size() || item.output_cols != V) {
            engine.RemoveQuery(qid);
            EngineSetError("perplexity: the engine returned no output tensor (set return_output_tensors = true)");
            return false;
        }
        const int rows = item.output_rows - 1;         // row i scores token i + 1
        std::vector<double> nll((size_t)host_threads, 0.0), nll2((size_t)host_threads, 0.0);
        auto work = [&](int w) {
            for (int i = w; i < rows; i += host_threads) {
                const double v = TokenNll(item.output_tensor.data() + (size_t)i * V, V, window[(size_t)i + 1]);
                nll[(size_t)w] += v; nll2[(size_t)w] += v * v;
            }
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < host_threads; w++) pool.emplace_back(work, w);
        work(0);
        for (std::thread &t : pool) t.join();
        for (int w = 0; w < host_threads; w++) { out.nll_sum += nll[(size_t)w]; out.nll2_sum += nll2[(size_t)w]; }
        out.count += rows;
        out.running.push_back(out.count > 0 ? exp(out.nll_sum / (double)out.count) : 0.0);
        engine.RemoveQuery(qid);
    }
    if (out.count < 2) { EngineSetError("perplexity: fewer than two scored tokens"); return false; }
    const double mean = out.nll_sum / (double)out.count;
    double var = out.nll2_sum / (double)out.count - mean * mean;
    out.ppl = exp(mean);
    out.ppl_stderr = var > 0 ? sqrt(var / (double)(out.count - 1)) * out.ppl : 0.0;
    return true;
}

} // namespace inferflow_amd
