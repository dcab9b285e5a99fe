// inferflow_service.h -- HTTP shell over the InferenceEngine facade: the token-id counterpart of the reference's service
// (src/service/inferflow_service.h / .cc: InferFlowServiceCore + InferFlowService).
//
// Same structure: ONE core thread loops Infer -> record each query's next token -> CommitInferenceResult (every active query
// advances per step: the reference's dynamic batching, inferflow_service.cc:60-129); request handlers AddQuery and collect the
// tokens of their query, whole or as a stream of chunks (:141-300); a URL containing "/chat/completions" switches the OpenAI-shaped
// response (:477-500); "/stat" (or {"header": {"fn": "get_stat"}}) reports the engine state (:389-475).
// Difference, by scope (tokenizer and templates are out of scope): requests carry TOKEN IDS, responses too.
//
//   native  : {"prompt_token_ids": [1, 15043, ...], "max_output_len": 64, "decoding_alg": "sample.top_p", "random_seed": 1,
//              "temperature": 0.8, "is_streaming_mode": false, "eos_token_id": 2}
//             -> {"ret_code": "succ", "token_ids": [...], "is_end": true, "time_cost": 0.123}     (streaming: one such object per chunk)
//   OpenAI  : POST /v1/chat/completions {"messages": [{"role": "user", "content_token_ids": [...]}], "max_tokens": 64,
//              "temperature": 0.8, "seed": 1, "stream": false}
//             -> {"id": "...", "object": "chat.completion", "choices": [{"index": 0, "message": {"role": "assistant", "token_ids": [...]},
//                 "finish_reason": "length" | "stop"}], "usage": {...}}      (stream: "chat.completion.chunk" objects with "delta")
#pragma once
#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "inference_engine.h"

namespace inferflow_amd {

struct InferFlowRequest {                 // InferFlowRequest (inferflow_service.h), token-id form
    std::vector<int> prompt_token_ids;
    int max_output_len = 64;
    std::string decoding_alg;
    int random_seed = 0;
    float temperature = 1.0f;
    bool is_streaming_mode = false;
    int eos_token_id = -1;
    std::string fn;                       // "" / "process_query" | "get_stat"
};

struct InferFlowResponseChunk {           // InferFlowResponseChunk
    std::string ret_code;
    std::vector<int> token_ids;
    bool is_end = false;
    float time_cost = 0;
    int prompt_tokens = 0;
    // why the query ended: "stop" (the request's eos_token_id was produced) | "length" (max_output_len, or the context is full)
    // | "error" (the engine step failed; ret_code says which).  The OpenAI shape reports it as finish_reason.
    std::string finish_reason;
    void ToJson(std::string &out) const;
    void ToJsonOpenAI(std::string &out, bool is_chunk, const std::string &id) const;
};

class InferFlowServiceCore {
public:
    explicit InferFlowServiceCore(QueryEngine &engine) : engine_(engine) {}
    ~InferFlowServiceCore() { Stop(); }
    void Start();                          // spawns the Infer loop
    void Stop();
    // JSON body -> request (native and OpenAI shapes); false: malformed
    static bool ParseRequest(InferFlowRequest &request, const std::string &body, bool is_openai_mode, std::string *err = nullptr);
    // Runs the query to its end.  on_chunk (may be null) is called with every batch of new tokens (streaming); returns the whole result.
    bool ProcessQuery(InferFlowResponseChunk &result, const InferFlowRequest &request,
                      const std::function<bool(const InferFlowResponseChunk &)> *on_chunk);
    void GetStat(std::string &json) const;

private:
    struct QueryResult { std::vector<int> tokens; bool is_end = false; int max_len = 0, eos = -1, produced = 0; std::string reason, err; };
    bool InferOnce();
    QueryEngine &engine_;
    std::thread loop_;
    std::atomic<bool> running_{false};
    mutable std::mutex engine_lock_;     // the facade's query table is not thread-safe: AddQuery / Infer + Commit / RemoveQuery take turns
    mutable std::mutex lock_;            // query_to_result_ (taken inside engine_lock_, never the other way round)
    std::map<int, QueryResult> query_to_result_;
    std::atomic<long long> steps_{0}, tokens_out_{0}, queries_{0}, retries_{0};      // retries_: steps run a second time after a failed Infer()
};

// Minimal HTTP/1.1 front (one thread per connection, at most MAX_CONNECTIONS of them, Connection: close; streaming responses use
// chunked transfer encoding).  Shutdown: RequestStop() only sets a flag (callable from a signal handler); Serve() polls the
// listening socket, sees the flag and returns; Stop() -- from the thread that owns the service, never from a handler -- stops the
// core loop (which releases every handler waiting in ProcessQuery) and waits until the last connection thread has left
// HandleConnection, so that the engine and the core outlive every user.
class InferFlowService {
public:
    static constexpr int MAX_CONNECTIONS = 256;
    InferFlowService(QueryEngine &engine) : core_(engine) {}
    ~InferFlowService() { Stop(); }
    bool Start(int port, int *bound_port = nullptr);       // port 0: any free port
    void RequestStop() { stop_.store(true); }               // async-signal-safe
    void Stop();
    void Serve();                                           // accept loop (blocks until RequestStop / Stop)
private:
    void HandleConnection(int fd);
    InferFlowServiceCore core_;
    int listen_fd_ = -1;
    std::atomic<bool> stop_{false};
    std::atomic<int> connections_{0};
};

} // namespace inferflow_amd
