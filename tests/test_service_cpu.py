"""CPU: the service shell's request parser and response formatter (host/inferflow_service.cc -- the token-id counterpart of
src/service/inferflow_service.cc:141-300, 477-570), through the two host-only C entry points."""
import ctypes as C
import json

import numpy as np

import inferflow_amd as ia


def _parse(body, openai=False):
    buf = C.create_string_buffer(1 << 16)
    rc = ia.lib().ifa_service_parse_request(body.encode(), int(openai), buf, len(buf))
    return rc, json.loads(buf.value.decode())


def test_native_request_fields():
    rc, r = _parse('{"prompt_token_ids": [1, 15043, 3186], "max_output_len": 40, "decoding_alg": "sample.top_p", "random_seed": 7, '
                   '"temperature": 0.75, "is_streaming_mode": true, "eos_token_id": 2}')
    assert rc == 0
    assert r["prompt_token_ids"] == [1, 15043, 3186] and r["max_output_len"] == 40 and r["decoding_alg"] == "sample.top_p"
    assert r["random_seed"] == 7 and abs(r["temperature"] - 0.75) < 1e-6 and r["is_streaming_mode"] is True and r["eos_token_id"] == 2


def test_openai_request_concatenates_the_messages_token_ids():
    rc, r = _parse('{"messages": [{"role": "system", "content_token_ids": [1, 5]}, {"role": "user", "content_token_ids": [9, 10, 11]}], '
                   '"max_tokens": 12, "seed": 3, "stream": false, "temperature": 0.5}', openai=True)
    assert rc == 0
    assert r["prompt_token_ids"] == [1, 5, 9, 10, 11] and r["max_output_len"] == 12 and r["random_seed"] == 3 and r["is_streaming_mode"] is False


def test_defaults_and_stat_function():
    rc, r = _parse('{"header": {"fn": "get_stat"}}')
    assert rc == 0 and r["fn"] == "get_stat" and r["prompt_token_ids"] == [] and r["max_output_len"] == 64 and r["eos_token_id"] == -1


def test_malformed_bodies_are_rejected():
    for body in ("not json", "[1, 2]", '{"prompt_token_ids": [1, {}]}', '{"prompt_token_ids": "1,2"}'):
        rc, r = _parse(body)
        assert rc == -1 and r["ret_code"] == "error.invalid_request_format", body
    rc, r = _parse('{"messages": [{"role": "user", "content": "text needs the tokenizer"}]}', openai=True)
    assert rc == -1 and r["ret_code"] == "error.invalid_request_format"


def _fmt(ids, is_end, openai, chunk, prompt_tokens=3):
    a = np.asarray(ids, np.int32)
    buf = C.create_string_buffer(1 << 16)
    rc = ia.lib().ifa_service_format_response(a.ctypes.data_as(C.POINTER(C.c_int)), len(a), int(is_end), int(openai), int(chunk), prompt_tokens, buf, len(buf))
    assert rc == 0
    return json.loads(buf.value.decode())


def test_response_shapes():
    r = _fmt([5, 6, 7], True, False, False)
    assert r["ret_code"] == "succ" and r["token_ids"] == [5, 6, 7] and r["is_end"] is True
    r = _fmt([5, 6, 7], True, True, False)
    assert r["object"] == "chat.completion" and r["choices"][0]["message"]["token_ids"] == [5, 6, 7]
    assert r["choices"][0]["finish_reason"] == "length" and r["usage"] == {"prompt_tokens": 3, "completion_tokens": 3, "total_tokens": 6}
    r = _fmt([8], False, True, True)
    assert r["object"] == "chat.completion.chunk" and r["choices"][0]["delta"]["token_ids"] == [8] and r["choices"][0]["finish_reason"] is None


# ---- the Infer / Commit loop over a host-only engine with InferenceEngine's query-table semantics (a query the
# engine ends by itself -- context full -- produces no item; the handler used to wait for it forever and leak its KV slot)
def _loop(max_ctx, prompt, max_output_len, eos=-1, n_requests=1, max_queries=2, fail_at=0, timeout_ms=3000):
    a = np.asarray(prompt, np.int32)
    buf = C.create_string_buffer(1 << 18)
    rc = ia.lib().ifa_service_selftest_loop(max_ctx, max_queries, fail_at, a.ctypes.data_as(C.POINTER(C.c_int)), len(a), max_output_len, eos,
                                            n_requests, timeout_ms, buf, len(buf))
    assert rc == 0
    return json.loads(buf.value.decode())


def test_request_longer_than_the_context_is_clamped_and_returns():
    # prompt 5 + max_output_len 100 > max_context_len 16: the output is cut at what fits, the handler returns, the slot goes back
    rs = _loop(16, [1, 2, 3, 4, 5], 100, n_requests=5)
    for r in rs:
        assert r["ok"] and not r["hung"] and r["ret_code"] == "succ" and r["is_end"] and r["finish_reason"] == "length", r
        assert r["token_ids"] == list(range(6, 6 + 11)) and r["active"] == 0, r       # room = 16 - 5: the engine's own bound (AddQuery / Infer)
    assert rs[0]["openai"]["choices"][0]["finish_reason"] == "length"


def test_unbounded_request_without_eos_ends_at_the_context_limit():
    # max_output_len <= 0 and no EOS: used to spin forever; after max_concurrent_queries of them every client got error.busy
    rs = _loop(12, [7, 8, 9], 0, n_requests=4, max_queries=2)
    for r in rs:
        assert r["ok"] and not r["hung"] and r["ret_code"] == "succ" and len(r["token_ids"]) == 9 and r["active"] == 0, r


def test_eos_ends_with_finish_reason_stop():
    rs = _loop(64, [10, 11], 30, eos=15)
    r = rs[0]
    assert r["ok"] and r["token_ids"] == [12, 13, 14, 15] and r["finish_reason"] == "stop" and r["openai"]["choices"][0]["finish_reason"] == "stop"


def test_prompt_that_leaves_no_room_is_refused():
    r = _loop(8, [1, 2, 3, 4, 5, 6, 7, 8], 4)[0]
    assert not r["ok"] and not r["hung"] and r["ret_code"] == "error.too_long_request" and r["active"] == 0
    # a prompt of max_ctx - 1 tokens is what AddQuery still accepts: one token comes back (the shell used to refuse it)
    r = _loop(8, [1, 2, 3, 4, 5, 6, 7], 4)[0]
    assert r["ok"] and not r["hung"] and r["ret_code"] == "succ" and r["token_ids"] == [8] and r["finish_reason"] == "length" and r["active"] == 0, r


def test_failed_engine_step_ends_the_query_with_an_error_and_frees_its_slot():
    # the 3rd Infer call AND its retry fail: the handler returns an error instead of spinning, the next requests are served
    rs = _loop(64, [1, 2], 20, n_requests=3, max_queries=1, fail_at=-3)
    assert not rs[0]["ok"] and not rs[0]["hung"] and rs[0]["ret_code"] == "error.inference_failed" and rs[0]["active"] == 0, rs[0]
    assert "error" in rs[0]["openai"]
    for r in rs[1:]:
        assert r["ok"] and r["ret_code"] == "succ" and len(r["token_ids"]) == 20, r


def test_a_step_that_fails_once_is_run_again():
    # a recoverable failure (the worker's bounded in-launch wait gave up: "repeat it: the waiting launches are off now"): the loop runs
    # the same step once more and no in-flight request is lost
    rs = _loop(64, [1, 2], 20, n_requests=3, max_queries=1, fail_at=3)
    for r in rs:
        assert r["ok"] and not r["hung"] and r["ret_code"] == "succ" and r["token_ids"] == list(range(3, 23)) and r["active"] == 0, r