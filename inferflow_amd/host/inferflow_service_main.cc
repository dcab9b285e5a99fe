// ifa_service -- the HTTP shell as a process (reference: src/service/inferflow_service_main.cc):
//   ifa_service <config.ini> [--section transformer_engine] [--port 8080]
// prints "listening on 127.0.0.1:<port>" once the engine is loaded; token-id requests, see inferflow_service.h
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <functional>

#include "inferflow_service.h"

using namespace inferflow_amd;

static InferFlowService *g_service = nullptr;
// the handler only sets a flag (async-signal-safe); Serve() sees it and returns, main() then stops the service in order
static void on_signal(int) { if (g_service) g_service->RequestStop(); }

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <config.ini> [--section S] [--port N]\n", argv[0]); return 2; }
    std::string ini = argv[1], section = "transformer_engine";
    int port = 8080;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--section") section = next();
        else if (a == "--port") port = atoi(next());
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    InferenceConfig cfg;
    if (!InferenceEngine::LoadConfig(cfg, ini, section)) { fprintf(stderr, "LoadConfig: %s\n", EngineLastError()); return 1; }
    InferenceEngine engine;
    if (!engine.Init(cfg)) { fprintf(stderr, "Init: %s\n", EngineLastError()); return 1; }
    InferFlowService service(engine);
    int bound = 0;
    if (!service.Start(port, &bound)) { fprintf(stderr, "cannot listen on port %d\n", port); return 1; }
    g_service = &service;
    signal(SIGINT, on_signal); signal(SIGTERM, on_signal);
    printf("listening on 127.0.0.1:%d\n", bound); fflush(stdout);
    service.Serve();
    service.Stop();          // core loop joined, every connection thread out of the service -- before `service` and `engine` are destroyed
    g_service = nullptr;
    return 0;
}
