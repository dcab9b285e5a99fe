#!/usr/bin/env python3
"""Decode rate at long contexts (Llama-2-7B Q4, F16 or Q8 KV): prefill n tokens, then 64 greedy steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth
kv = dt.Q8_B32T2 if "q8" in sys.argv else dt.F16
ctxs = [int(v) for v in os.environ.get("IFA_CTX", "1024,4096,16384").split(",")]
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, kv, max_ctx=max(ctxs) + 80)
rng = np.random.default_rng(1)
for n in ctxs:
    wk.reset()
    pr = rng.integers(3, s["vocab"], n).astype(np.int32)
    tok = None
    torch.cuda.synchronize(); tp0 = time.perf_counter()
    for c0 in range(0, n, 2048):                      # prefill in chunks of 2048 tokens
        tok = wk.forward(pr[c0:c0 + 2048], c0)
    torch.cuda.synchronize(); tp = time.perf_counter() - tp0
    wk.decode(tok, n, 8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks, ms = wk.decode(tok, n + 8, 64)
    torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
    print("context %6d  kv %s  prefill %.0f tok/s (%.2f s, chunks of 2048)  decode %.1f tok/s (%.3f ms/token)" % (
        n, dt.name(kv), n / tp, tp, 64 / dt_s, dt_s * 1e3 / 64), flush=True)
