"""-m gpu: the HTTP service shell end to end (bin/ifa_service over a small model): native and OpenAI-shaped requests, a streamed
response, concurrent clients (the core's Infer loop advances every active query per step) and /stat; the greedy token ids must be
those of the InferenceEngine facade on the same .ini.  Reference: src/service/inferflow_service.cc:60-129, 141-300, 477-570."""
import http.client
import json
import os
import subprocess
import threading

import numpy as np
import pytest

from inferflow_amd import build
from inferflow_amd.engine import InferenceEngine
from tests import engine_fixtures as fx

pytestmark = pytest.mark.gpu


def _post(port, url, body, timeout=60):
    c = http.client.HTTPConnection("127.0.0.1", port, timeout=timeout)
    c.request("POST", url, body=json.dumps(body), headers={"Content-Type": "application/json"})
    r = c.getresponse()
    data = r.read().decode()
    c.close()
    return r.status, data


def test_service_shell_serves_token_id_queries(tmp_path):
    build.build_library()
    ini, _ = fx.write_model_dir(str(tmp_path / "m"), fmt="llama2.c", wd="Q4", kvd="F16", ctx=128, ret="false", maxq=4)
    prompt = [int(t) for t in np.random.default_rng(5).integers(3, 900, 9)]
    eng = InferenceEngine.from_ini(ini)
    qid = eng.add_query(np.asarray(prompt, np.int32))
    want, _ = eng.generate(qid, 12)
    want = [int(t) for t in want]
    eng.close()
    exe = os.path.join(build.BIN_DIR, "ifa_service")
    p = subprocess.Popen([exe, ini, "--port", "0"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        line = p.stdout.readline()
        assert line.startswith("listening on 127.0.0.1:"), line
        port = int(line.strip().rsplit(":", 1)[1])
        # native request
        st, data = _post(port, "/", {"prompt_token_ids": prompt, "max_output_len": 12, "decoding_alg": "greedy"})
        r = json.loads(data)
        assert st == 200 and r["ret_code"] == "succ" and r["is_end"] is True and r["token_ids"] == want, (r, want)
        # OpenAI-shaped request
        st, data = _post(port, "/v1/chat/completions", {"messages": [{"role": "user", "content_token_ids": prompt}], "max_tokens": 12})
        r = json.loads(data)
        assert st == 200 and r["object"] == "chat.completion" and r["choices"][0]["message"]["token_ids"] == want
        assert r["usage"]["prompt_tokens"] == len(prompt) and r["choices"][0]["finish_reason"] == "length"
        # streamed (chunked transfer): the chunks concatenate to the same ids, the last one says is_end
        st, data = _post(port, "/", {"prompt_token_ids": prompt, "max_output_len": 12, "is_streaming_mode": True})
        chunks = [json.loads(c) for c in data.split("\n\n") if c.strip()]
        assert st == 200 and sum((c["token_ids"] for c in chunks), []) == want and chunks[-1]["is_end"] is True
        # four clients at once: every one gets the single-query answer (each query has its own KV slot, the loop batches them)
        out = [None] * 4
        def go(i):
            out[i] = json.loads(_post(port, "/", {"prompt_token_ids": prompt, "max_output_len": 12})[1])
        ts = [threading.Thread(target=go, args=(i,)) for i in range(4)]
        [t.start() for t in ts]; [t.join() for t in ts]
        for r in out:
            assert r["ret_code"] == "succ" and len(r["token_ids"]) == 12
        assert sum(r["token_ids"] == want for r in out) >= 3       # (a batched step runs T > 1 kernels: F16 activations may flip a near-tie)
        # more output than the context holds (prompt 9 + 500 > max_context_len 128): cut at what fits, the handler returns and
        # the KV slot goes back -- asked maxq + 1 times, so a leaked slot would answer error.busy
        for _ in range(5):
            st, data = _post(port, "/", {"prompt_token_ids": prompt, "max_output_len": 500})
            r = json.loads(data)
            assert st == 200 and r["ret_code"] == "succ" and r["is_end"] is True and len(r["token_ids"]) == 128 - len(prompt), r      # the engine's own bound: a step needs prompt + produced < max_ctx
            assert r["token_ids"][:12] == want
        # EOS: the third token of the greedy continuation as eos_token_id ends the query there, finish_reason "stop"
        eos = want[2]
        k = want.index(eos) + 1
        st, data = _post(port, "/v1/chat/completions", {"messages": [{"role": "user", "content_token_ids": prompt}], "max_tokens": 12, "eos_token_id": eos})
        r = json.loads(data)
        assert st == 200 and r["choices"][0]["message"]["token_ids"] == want[:k] and r["choices"][0]["finish_reason"] == "stop", r
        # an empty request is refused, the engine keeps serving
        st, data = _post(port, "/", {"prompt_token_ids": []})
        assert st == 400 and json.loads(data)["ret_code"] == "error.empty_request"
        c = http.client.HTTPConnection("127.0.0.1", port, timeout=10)
        c.request("GET", "/stat")
        stat = json.loads(c.getresponse().read().decode())
        c.close()
        assert stat["active_queries"] == 0 and stat["served_queries"] >= 13 and stat["output_tokens"] >= 7 * 12
    finally:
        p.terminate()
        p.wait(timeout=20)
