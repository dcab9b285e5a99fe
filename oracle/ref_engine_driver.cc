// ref_engine_driver.cc -- TEST INFRASTRUCTURE (oracle/_ref): drives the reference's own InferenceEngine (CPU build,
// compiled from the sources where they lie under /root/reference by oracle/Makefile's `ref_engine` target -- plain
// g++/gcc command lines, not the reference's CMake) through its public interface
//   InferenceEngine::LoadConfig / Init / AddQuery(tokens) / Infer / CommitInferenceResult
//   (src/transformer/inference_engine.h:32-129, used as src/tools/llm_inference.cc:183-414 uses it)
// and dumps what tests/golden/gen_model_fixtures.py needs: the full logits of every step (return_output_tensors) and the
// greedy token ids.  It can also time the reference CPU path (bench.py's cpu_baseline leg, kind "reference").
// Nothing in inferflow_amd/ links or loads this.
//
//   ifa_ref_engine <engine.ini> <prompt_ids.i32> <n_steps> <out.bin> [quiet]
// out.bin: int32 magic 0x49464131, int32 vocab, int32 prompt_len, int32 n_steps, then
//   float32 [prompt_len][vocab] logits of the prefill step, int32 token,
//   (float32 [vocab] logits, int32 token) per further step, then float64 prefill_ms, float64 decode_ms_total.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>
#include "sslib/app_environment.h"
#include "transformer/inference_engine.h"

using namespace inferflow;
using namespace inferflow::transformer;

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool dump_logits(FILE *f, const HostTensor &t, int vocab, int rows_expected)
{
    const int cols = t.ne[0], rows = t.ne[1] > 0 ? t.ne[1] : 1;
    if (cols != vocab || rows != rows_expected) {
        fprintf(stderr, "output tensor is %d x %d, expected %d x %d\n", rows, cols, rows_expected, vocab);
        return false;
    }
    std::vector<float> buf((size_t)rows * cols);
    if (t.data_type == ElementType::F32) {
        const float *p = t.data_f32();
        for (size_t i = 0; i < buf.size(); i++) buf[i] = p[i];
    } else if (t.data_type == ElementType::F16) {
        const inferflow_fp16 *p = t.data_f16();
        for (size_t i = 0; i < buf.size(); i++) buf[i] = (float)p[i];
    } else {
        fprintf(stderr, "unexpected output tensor type %d\n", (int)t.data_type);
        return false;
    }
    return fwrite(buf.data(), sizeof(float), buf.size(), f) == buf.size();
}

int main(int argc, const char *argv[])
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <engine.ini> <prompt_ids.i32> <n_steps> <out.bin> [quiet]\n", argv[0]);
        return 2;
    }
    const std::string ini = argv[1];
    const int n_steps = atoi(argv[3]);
    const bool quiet = argc > 5;
    std::vector<int> prompt;
    {
        FILE *pf = fopen(argv[2], "rb");
        if (!pf) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
        int32_t v;
        while (fread(&v, 4, 1, pf) == 1) prompt.push_back(v);
        fclose(pf);
    }
    if (!sslib::InitAppEnv(ini, "ifa_ref_engine", "0.1.0")) { fprintf(stderr, "InitAppEnv failed\n"); return 3; }

    InferenceConfig cfg;
    if (!InferenceEngine::LoadConfig(cfg, ini, "transformer_engine")) { fprintf(stderr, "LoadConfig failed\n"); return 3; }
    cfg.data_dir = sslib::AppEnv::DataRootDir();
    InferenceEngine engine;
    if (!engine.Init(cfg)) { fprintf(stderr, "Init failed\n"); return 3; }
    const int vocab = engine.vocabulary().Size();
    FILE *out = fopen(argv[4], "wb");
    if (!out) { fprintf(stderr, "cannot open %s\n", argv[4]); return 2; }
    const int32_t hdr[4] = {0x49464131, vocab, (int32_t)prompt.size(), n_steps};
    fwrite(hdr, 4, 4, out);

    SamplingStrategy::QueryOptions opts;
    opts.strategy_id = SamplingStrategyId::Greedy;
    const int qid = engine.AddQuery(prompt, opts);
    if (qid <= 0) { fprintf(stderr, "AddQuery returned %d\n", qid); return 4; }

    double prefill_ms = 0.0, decode_ms = 0.0;
    InferenceResult res;
    for (int step = 0; step < n_steps; step++) {
        const double t0 = now_ms();
        if (!engine.Infer(res) || res.items.size() != 1) { fprintf(stderr, "Infer failed at step %d\n", step); return 5; }
        const double dt = now_ms() - t0;
        if (step == 0) prefill_ms = dt; else decode_ms += dt;
        const QueryInferenceResult &qr = *res.items[0];
        const int32_t tok = qr.next_tokens[0].id;
        if (!quiet && !dump_logits(out, qr.output_tensor, vocab, step == 0 ? (int)prompt.size() : 1)) return 6;
        fwrite(&tok, 4, 1, out);
        std::map<int, QueryNextToken> commit;
        QueryNextToken nt; nt.id = tok; nt.is_end = false;
        commit[qid] = nt;
        if (!engine.CommitInferenceResult(commit)) { fprintf(stderr, "Commit failed at step %d\n", step); return 7; }
    }
    fwrite(&prefill_ms, 8, 1, out);
    fwrite(&decode_ms, 8, 1, out);
    fclose(out);
    fprintf(stderr, "ref engine: vocab %d, prompt %d, steps %d, prefill %.2f ms, decode %.3f ms/token\n", vocab,
            (int)prompt.size(), n_steps, prefill_ms, n_steps > 1 ? decode_ms / (n_steps - 1) : 0.0);
    return 0;
}
