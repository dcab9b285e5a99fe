// ifa_perplexity -- token-id counterpart of the reference's perplexity tool (src/tools/perplexity.cc:165-300,
// bin/perplexity.ini): [main] inference_engine_config / test_data_file / max_length / stride.
//   ifa_perplexity <perplexity.ini> [--section transformer_engine]
// test_data_file holds whitespace- or comma-separated token ids (tokenizers are outside the hot path).
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

#include "ifa_ini.h"
#include "perplexity.h"

using namespace inferflow_amd;

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <perplexity.ini> [--section S]\n", argv[0]); return 2; }
    std::string section = "transformer_engine";
    for (int i = 2; i + 1 < argc; i++) if (std::string(argv[i]) == "--section") section = argv[++i];
    IniConfig ini; std::string err;
    if (!ini.Load(argv[1], &err)) { fprintf(stderr, "Failed to load the configuration data: %s\n", err.c_str()); return 1; }
    std::string engine_ini, data_file; int max_length = 512, stride = 512;
    if (!ini.GetItem("main", "inference_engine_config", engine_ini) || !ini.GetItem("main", "test_data_file", data_file)) {
        fprintf(stderr, "[main] needs inference_engine_config and test_data_file\n"); return 1;
    }
    ini.GetItem("main", "max_length", max_length);
    ini.GetItem("main", "stride", stride);

    InferenceConfig cfg;
    if (!InferenceEngine::LoadConfig(cfg, engine_ini, section)) { fprintf(stderr, "Failed to load the inference configuration: %s\n", EngineLastError()); return 1; }
    cfg.max_concurrent_queries = 1;            // perplexity.cc:176
    cfg.return_output_tensors = true;
    InferenceEngine engine;
    if (!engine.Init(cfg)) { fprintf(stderr, "Failed to initialize the inference engine: %s\n", EngineLastError()); return 1; }

    std::ifstream f(data_file);
    if (!f) { fprintf(stderr, "Failed to open the file: %s\n", data_file.c_str()); return 1; }
    std::stringstream ss; ss << f.rdbuf();
    std::string text = ss.str();
    for (char &c : text) if (c == ',') c = ' ';
    std::istringstream in(text);
    std::vector<int> tokens; long long v;
    while (in >> v) tokens.push_back((int)v);

    PerplexityResult r;
    if (!ComputePerplexity(engine, tokens, max_length, stride, r)) { fprintf(stderr, "perplexity: %s\n", EngineLastError()); return 1; }
    for (size_t i = 0; i < r.running.size(); i++) printf("[%zu]%.4lf\n", i, r.running[i]);
    printf("Final estimate: PPL = %.4lf +/- %.5lf\n", r.ppl, r.ppl_stderr);
    return 0;
}
