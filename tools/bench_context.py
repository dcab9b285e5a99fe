#!/usr/bin/env python3
"""Decode rate vs context length (Llama-2-7B Q4): prefill N tokens, then time 64 decode steps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
kv = dt.Q8_B32T2 if "q8" in sys.argv else dt.F16
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, kv, max_ctx=4200)
rng = np.random.default_rng(1)
for n in (16, 256, 1024, 2048, 4096):
    pr = rng.integers(3, s["vocab"], n).astype(np.int32)
    tok = wk.forward(pr, 0)
    wk.decode(tok, n, 8)
    toks, ms = wk.decode(tok, n, 64)
    attn_us = wk.time_kernel(1, 50)
    print(json.dumps({"context": n, "kv": dt.name(kv), "decode_tok_s": 64e3 / ms, "attn_kernel_us": attn_us}), flush=True)
