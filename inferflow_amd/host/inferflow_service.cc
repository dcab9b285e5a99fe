// inferflow_service.cc -- see inferflow_service.h (reference: src/service/inferflow_service.cc:60-129, 141-300, 337-475, 477-570)
#include <functional>
#include "inferflow_service.h"

#include <arpa/inet.h>
#include <poll.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <sstream>

#include "ifa_json.h"

namespace inferflow_amd {

static std::string JoinIds(const std::vector<int> &ids)
{
    std::string s = "[";
    for (size_t i = 0; i < ids.size(); i++) { if (i) s += ", "; s += std::to_string(ids[i]); }
    return s + "]";
}

void InferFlowResponseChunk::ToJson(std::string &out) const
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.4f", time_cost);
    out = "{\"ret_code\": \"" + ret_code + "\", \"token_ids\": " + JoinIds(token_ids) + ", \"is_end\": " + (is_end ? "true" : "false")
        + ", \"time_cost\": " + buf + "}";
}

void InferFlowResponseChunk::ToJsonOpenAI(std::string &out, bool is_chunk, const std::string &id) const
{
    const bool failed = !ret_code.empty() && ret_code != "succ";
    if (failed) { out = "{\"error\": {\"message\": \"" + ret_code + "\", \"type\": \"invalid_request_error\"}}"; return; }
    const std::string finish = !is_end ? "null" : ("\"" + (finish_reason.empty() ? std::string("length") : finish_reason) + "\"");
    out = "{\"id\": \"" + id + "\", \"object\": \"" + (is_chunk ? "chat.completion.chunk" : "chat.completion") + "\", \"choices\": [{\"index\": 0, \""
        + (is_chunk ? "delta" : "message") + "\": {\"role\": \"assistant\", \"token_ids\": " + JoinIds(token_ids) + "}, \"finish_reason\": " + finish + "}]";
    if (!is_chunk) out += ", \"usage\": {\"prompt_tokens\": " + std::to_string(prompt_tokens) + ", \"completion_tokens\": " + std::to_string(token_ids.size())
        + ", \"total_tokens\": " + std::to_string(prompt_tokens + (int)token_ids.size()) + "}";
    out += "}";
}

// all or nothing: a list with a non-number in it is refused, never half-read
static bool ReadIds(const JsonValue *v, std::vector<int> &out)
{
    if (!v || v->type != JsonValue::Array) return false;
    for (const JsonValue &e : v->arr) if (e.type != JsonValue::Number) return false;
    for (const JsonValue &e : v->arr) out.push_back((int)e.num);
    return true;
}

bool InferFlowServiceCore::ParseRequest(InferFlowRequest &r, const std::string &body, bool is_openai_mode, std::string *err)
{
    JsonValue root; JsonParser parser; std::string perr;
    if (!parser.Parse(body, root, &perr) || root.type != JsonValue::Object) { if (err) *err = "error.invalid_request_format"; return false; }
    if (const JsonValue *h = root.Get("header")) h->GetString("fn", r.fn);
    if (is_openai_mode) {
        // the prompt: the token ids of every message in order (the chat template is applied by the caller: tokenizer out of scope)
        if (const JsonValue *msgs = root.Get("messages")) {
            if (msgs->type != JsonValue::Array) { if (err) *err = "error.invalid_request_format"; return false; }
            for (const JsonValue &m : msgs->arr)
                if (!ReadIds(m.Get("content_token_ids"), r.prompt_token_ids)) { if (err) *err = "error.invalid_request_format"; return false; }
        } else if (root.Get("prompt_token_ids") && !ReadIds(root.Get("prompt_token_ids"), r.prompt_token_ids)) { if (err) *err = "error.invalid_request_format"; return false; }
        root.GetNumber("max_tokens", r.max_output_len);
        root.GetNumber("seed", r.random_seed);
        root.GetBool("stream", r.is_streaming_mode);
    } else {
        if (root.Get("prompt_token_ids") && !ReadIds(root.Get("prompt_token_ids"), r.prompt_token_ids)) { if (err) *err = "error.invalid_request_format"; return false; }
        root.GetNumber("max_output_len", r.max_output_len);
        root.GetNumber("random_seed", r.random_seed);
        root.GetBool("is_streaming_mode", r.is_streaming_mode);
    }
    root.GetString("decoding_alg", r.decoding_alg);
    root.GetNumber("temperature", r.temperature);
    root.GetNumber("eos_token_id", r.eos_token_id);
    return true;
}

void InferFlowServiceCore::Start()
{
    if (running_.exchange(true)) return;
    loop_ = std::thread([this] { while (running_.load()) InferOnce(); });
}

void InferFlowServiceCore::Stop()
{
    if (!running_.exchange(false)) return;
    if (loop_.joinable()) loop_.join();
}

// InferFlowServiceCore::Infer (inferflow_service.cc:73-131): one engine step for every active query.
// Every registered query leaves a step in one of three states: it got a token (recorded, committed, ended by the service's own
// EOS / max_len rule), the engine ended it without a token (its context is full: QueryEnded), or the step failed (Infer returned
// false: every query that got nothing ends with an error code) -- a handler never waits for a query the engine will not advance.
bool InferFlowServiceCore::InferOnce()
{
    std::unique_lock<std::mutex> eg(engine_lock_);
    if (engine_.QueryCount() == 0) { eg.unlock(); std::this_thread::sleep_for(std::chrono::milliseconds(1)); return true; }
    InferenceResult result;
    bool ok = engine_.Infer(result);
    // One retry of a step that failed as a whole: the worker's bounded in-launch waits report a timeout as a FAILED call
    // "repeat it: the waiting launches are off now" (IFA_ERR_STATE, include/inferflow_amd.h) -- recoverable by design, and a failed
    // Infer() commits nothing (Query::processed moves only behind a successful step), so the same step is simply run again; a second
    // failure ends the queries as before.
    if (!ok && result.items.empty()) { retries_++; ok = engine_.Infer(result); }
    if (!result.items.empty()) steps_++;
    std::map<int, QueryNextToken> commit;
    {
        std::lock_guard<std::mutex> g(lock_);
        for (const QueryInferenceResult &item : result.items) {
            if (item.next_tokens.empty()) continue;
            auto it = query_to_result_.find(item.query_id);
            if (it == query_to_result_.end()) continue;          // (a query the handler already dropped)
            const int id = item.next_tokens[0].id;
            QueryResult &qr = it->second;
            qr.tokens.push_back(id);
            qr.produced++;
            tokens_out_++;
            QueryNextToken nt; nt.id = id;
            const bool eos = qr.eos >= 0 && id == qr.eos;
            nt.is_end = eos || (qr.max_len > 0 && qr.produced >= qr.max_len);
            if (nt.is_end) { qr.is_end = true; qr.reason = eos ? "stop" : "length"; }
            commit[item.query_id] = nt;
        }
    }
    if (!commit.empty()) engine_.CommitInferenceResult(commit);
    {
        std::lock_guard<std::mutex> g(lock_);
        for (auto &kv : query_to_result_) {
            QueryResult &qr = kv.second;
            if (qr.is_end || commit.count(kv.first)) continue;
            if (!ok) { qr.is_end = true; qr.reason = "error"; qr.err = "error.inference_failed"; }
            else if (engine_.QueryEnded(kv.first)) { qr.is_end = true; qr.reason = "length"; }      // context full (or the query is gone)
        }
    }
    if (result.items.empty()) { eg.unlock(); std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
    return ok;
}

bool InferFlowServiceCore::ProcessQuery(InferFlowResponseChunk &result, const InferFlowRequest &request,
                                        const std::function<bool(const InferFlowResponseChunk &)> *on_chunk)
{
    const auto t0 = std::chrono::steady_clock::now();
    result = InferFlowResponseChunk();
    result.prompt_tokens = (int)request.prompt_token_ids.size();
    if (request.prompt_token_ids.empty()) { result.ret_code = "error.empty_request"; return false; }
    // the output length is bounded by the context whatever the request says (0 or negative: "as much as fits"): a query the
    // engine ends by itself produces no token, so an unbounded request would otherwise only end on EOS
    const int max_ctx = engine_.MaxContextLen();
    // (a prompt of P tokens yields at most max_ctx - P tokens: the step that produces token k + 1 needs P + k < max_ctx, the same
    //  bound as InferenceEngine::AddQuery / Infer -- a prompt of max_ctx - 1 tokens is accepted and yields one token)
    const int room = max_ctx - (int)request.prompt_token_ids.size();
    if (room < 1) { result.ret_code = "error.too_long_request"; return false; }
    const int max_len = request.max_output_len > 0 ? std::min(request.max_output_len, room) : room;
    QueryOptions qo;
    qo.strategy_id = (int)engine_.GetSamplingStrategyId(request.decoding_alg);
    qo.random_seed = request.random_seed;
    qo.temperature = request.temperature;
    qo.max_output_len = max_len;
    int qid = 0;
    {
        // registered before the loop can step the query (the loop holds engine_lock_ for a whole Infer + Commit)
        std::lock_guard<std::mutex> eg(engine_lock_);
        std::lock_guard<std::mutex> g(lock_);
        qid = engine_.AddQuery(request.prompt_token_ids, qo);
        if (qid > 0) { QueryResult &qr = query_to_result_[qid]; qr.max_len = max_len; qr.eos = request.eos_token_id; }
    }
    if (qid <= 0) { result.ret_code = qid == 0 ? "error.busy" : "error.invalid_query"; return false; }
    queries_++;
    bool is_end = false;
    std::string reason, err;
    while (!is_end && running_.load()) {
        std::this_thread::sleep_for(std::chrono::microseconds(500));
        std::vector<int> fresh;
        {
            std::lock_guard<std::mutex> g(lock_);
            auto it = query_to_result_.find(qid);
            if (it == query_to_result_.end()) break;
            fresh.swap(it->second.tokens);
            is_end = it->second.is_end;
            if (is_end) { reason = it->second.reason; err = it->second.err; query_to_result_.erase(it); }
        }
        result.token_ids.insert(result.token_ids.end(), fresh.begin(), fresh.end());
        if (on_chunk && (!fresh.empty() || is_end)) {
            InferFlowResponseChunk chunk;
            chunk.token_ids = fresh; chunk.is_end = is_end; chunk.finish_reason = reason;
            chunk.time_cost = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
            if (!(*on_chunk)(chunk)) {      // the client went away: drop the query (reference: engine_.RemoveQuery on a failed WriteChunk)
                std::lock_guard<std::mutex> eg(engine_lock_);
                engine_.RemoveQuery(qid);
                std::lock_guard<std::mutex> g(lock_);
                query_to_result_.erase(qid);
                return false;
            }
        }
    }
    {   // the query is done (or the service stops): its slot and KV cache go back to the engine
        std::lock_guard<std::mutex> eg(engine_lock_);
        engine_.RemoveQuery(qid);
        std::lock_guard<std::mutex> g(lock_);
        query_to_result_.erase(qid);
    }
    result.is_end = is_end;
    result.finish_reason = reason;
    result.time_cost = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
    if (!err.empty()) { result.ret_code = err; return false; }
    result.ret_code = is_end ? "succ" : "error.service_stopped";
    return is_end;
}

void InferFlowServiceCore::GetStat(std::string &json) const
{
    std::lock_guard<std::mutex> eg(engine_lock_);
    json = "{\"version\": \"" + engine_.Version() + "\", \"model\": \"" + engine_.ModelId() + "\", \"active_queries\": " + std::to_string(engine_.QueryCount())
        + ", \"served_queries\": " + std::to_string(queries_.load()) + ", \"engine_steps\": " + std::to_string(steps_.load())
        + ", \"output_tokens\": " + std::to_string(tokens_out_.load()) + ", \"vocab_size\": " + std::to_string(engine_.VocabSize()) + "}";
}

// ------------------------------------------------------------------ HTTP front
static bool SendAll(int fd, const std::string &s)
{
    size_t off = 0;
    while (off < s.size()) {
        const ssize_t n = send(fd, s.data() + off, s.size() - off, MSG_NOSIGNAL);
        if (n <= 0) return false;
        off += (size_t)n;
    }
    return true;
}

static std::string HttpHeader(int status, const std::string &content_type, long long body_len, bool chunked)
{
    std::string h = "HTTP/1.1 " + std::to_string(status) + (status == 200 ? " OK" : status == 400 ? " Bad Request" : status == 404 ? " Not Found" : status == 503 ? " Service Unavailable" : " Error") + "\r\n";
    h += "Content-Type: " + content_type + "\r\nConnection: close\r\n";
    if (chunked) h += "Transfer-Encoding: chunked\r\n";
    else h += "Content-Length: " + std::to_string(body_len) + "\r\n";
    return h + "\r\n";
}

bool InferFlowService::Start(int port, int *bound_port)
{
    listen_fd_ = socket(AF_INET, SOCK_STREAM, 0);
    if (listen_fd_ < 0) return false;
    int one = 1;
    setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in addr; memset(&addr, 0, sizeof addr);
    addr.sin_family = AF_INET; addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK); addr.sin_port = htons((uint16_t)port);
    if (bind(listen_fd_, (sockaddr *)&addr, sizeof addr) != 0 || listen(listen_fd_, 64) != 0) { close(listen_fd_); listen_fd_ = -1; return false; }
    socklen_t len = sizeof addr;
    if (bound_port && getsockname(listen_fd_, (sockaddr *)&addr, &len) == 0) *bound_port = ntohs(addr.sin_port);
    core_.Start();
    return true;
}

void InferFlowService::Stop()
{
    stop_.store(true);
    if (listen_fd_ >= 0) { shutdown(listen_fd_, SHUT_RDWR); close(listen_fd_); listen_fd_ = -1; }
    core_.Stop();        // running_ = false: every handler inside ProcessQuery leaves its wait loop and returns its query to the engine
    // the connection threads are detached but counted: nobody may still be inside HandleConnection (touching core_ / the engine)
    // when the owner destroys them
    for (int i = 0; i < 20000 && connections_.load() > 0; i++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
}

void InferFlowService::Serve()
{
    while (!stop_.load()) {
        pollfd pf; pf.fd = listen_fd_; pf.events = POLLIN; pf.revents = 0;
        const int pr = poll(&pf, 1, 100);            // the stop flag (set by a signal handler) is seen within 100 ms
        if (pr <= 0) continue;
        const int fd = accept(listen_fd_, nullptr, nullptr);
        if (fd < 0) { if (stop_.load()) break; continue; }
        if (connections_.load() >= MAX_CONNECTIONS) {
            const std::string b = "{\"ret_code\": \"error.busy\"}";
            SendAll(fd, HttpHeader(503, "application/json", (long long)b.size(), false) + b);
            close(fd);
            continue;
        }
        connections_++;
        std::thread([this, fd] { HandleConnection(fd); close(fd); connections_--; }).detach();
    }
}

// InferFlowService::HandleRequest (inferflow_service.cc:477-570)
void InferFlowService::HandleConnection(int fd)
{
    constexpr size_t MAX_REQUEST_LEN = 4u << 20;
    std::string buf;
    size_t hdr_end = std::string::npos;
    char tmp[8192];
    while ((hdr_end = buf.find("\r\n\r\n")) == std::string::npos && buf.size() < MAX_REQUEST_LEN) {
        const ssize_t n = recv(fd, tmp, sizeof tmp, 0);
        if (n <= 0) return;
        buf.append(tmp, (size_t)n);
    }
    if (hdr_end == std::string::npos) return;
    std::istringstream first(buf.substr(0, buf.find("\r\n")));
    std::string method, url;
    first >> method >> url;
    size_t content_len = 0;
    {
        std::string lower = buf.substr(0, hdr_end);
        for (char &c : lower) c = (char)tolower((unsigned char)c);
        const size_t p = lower.find("content-length:");
        if (p != std::string::npos) content_len = (size_t)atoll(lower.c_str() + p + 15);
    }
    if (content_len > MAX_REQUEST_LEN) { std::string b = "{\"ret_code\": \"error.too_long_request\"}"; SendAll(fd, HttpHeader(400, "application/json", (long long)b.size(), false) + b); return; }
    std::string body = buf.substr(hdr_end + 4);
    while (body.size() < content_len) {
        const ssize_t n = recv(fd, tmp, sizeof tmp, 0);
        if (n <= 0) return;
        body.append(tmp, (size_t)n);
    }
    const bool is_openai_mode = url.find("/chat/completions") != std::string::npos;
    if (method != "GET" && method != "POST" && method != "PUT") { SendAll(fd, HttpHeader(501, "application/json", 0, false)); return; }
    InferFlowRequest request;
    std::string perr;
    const bool has_body = !body.empty();
    if (has_body && !InferFlowServiceCore::ParseRequest(request, body, is_openai_mode, &perr)) {
        std::string b = "{\"ret_code\": \"" + perr + "\"}";
        SendAll(fd, HttpHeader(400, "application/json", (long long)b.size(), false) + b);
        return;
    }
    if (url.find("/stat") != std::string::npos || request.fn == "get_stat" || (method == "GET" && !has_body)) {
        std::string b; core_.GetStat(b);
        SendAll(fd, HttpHeader(200, "application/json", (long long)b.size(), false) + b);
        return;
    }
    static std::atomic<long long> serial{0};
    const std::string id = "ifa-" + std::to_string(++serial);
    InferFlowResponseChunk result;
    if (request.is_streaming_mode) {
        if (!SendAll(fd, HttpHeader(200, is_openai_mode ? "text/event-stream" : "application/json", 0, true))) return;
        auto write_chunk = [&](const std::string &payload) {
            char sz[32]; snprintf(sz, sizeof sz, "%zx\r\n", payload.size());
            return SendAll(fd, std::string(sz) + payload + "\r\n");
        };
        std::function<bool(const InferFlowResponseChunk &)> on_chunk = [&](const InferFlowResponseChunk &c) {
            std::string js;
            if (is_openai_mode) { c.ToJsonOpenAI(js, true, id); js = "data: " + js; }
            else { InferFlowResponseChunk cc = c; cc.ret_code = c.is_end ? "succ" : ""; cc.ToJson(js); }
            return write_chunk(js + "\n\n");
        };
        const bool ok = core_.ProcessQuery(result, request, &on_chunk);
        if (!ok && result.ret_code != "succ") { std::string js; if (is_openai_mode) result.ToJsonOpenAI(js, true, id); else result.ToJson(js); write_chunk(js + "\n\n"); }
        if (is_openai_mode) write_chunk("data: [DONE]\n\n");
        SendAll(fd, "0\r\n\r\n");
        return;
    }
    const bool ok = core_.ProcessQuery(result, request, nullptr);
    std::string js;
    if (is_openai_mode) result.ToJsonOpenAI(js, false, id); else result.ToJson(js);
    SendAll(fd, HttpHeader(ok ? 200 : 400, "application/json", (long long)js.size(), false) + js);
}

} // namespace inferflow_amd
