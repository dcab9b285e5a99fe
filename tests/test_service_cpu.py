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
