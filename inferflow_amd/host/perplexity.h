// perplexity.h -- the reference's perplexity harness (src/tools/perplexity.cc:41-284) over the
// InferenceEngine facade: the token stream is cut into windows [start, start + max_length) every
// `stride` tokens (LoadQueryList, :41-83), every window is ONE query whose whole-prompt logits come
// back through return_output_tensors, row i scores token i+1 with a float log-softmax
// (log_softmax, :100-119), and the totals give PPL = exp(mean nll) with the reference's error
// estimate (:268-276).  Tokenizers are outside the hot path: the stream is token ids.
#pragma once
#include <vector>

#include "inference_engine.h"

namespace inferflow_amd {

struct PerplexityResult {
    double ppl = 0, ppl_stderr = 0;     // "Final estimate: PPL = %.4lf +/- %.5lf"
    double nll_sum = 0, nll2_sum = 0;   // Σ -log p, Σ (log p)²
    long long count = 0;                // scored tokens (rows - 1 per window)
    std::vector<double> running;        // exp(nll/count) after each window ("[i]%.4lf" lines)
};

// -log softmax(logits)[token_id] of one F16 logits row, arithmetic of perplexity.cc:100-119
double TokenNll(const uint16_t *logits_f16, int vocab, int token_id);

// The engine must have been initialised with return_output_tensors = true and have no active query.
bool ComputePerplexity(InferenceEngine &engine, const std::vector<int> &tokens, int max_length, int stride,
                       PerplexityResult &out, int host_threads = 8);

} // namespace inferflow_amd
