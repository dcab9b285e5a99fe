#!/usr/bin/env python3
"""Decode rate by context with the default options (16-step calls, best of 3): the table of DESIGN.md "Decode by context".
    ctx_sweep.py [q8] [shape]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
kv = dt.Q8_B32T2 if "q8" in sys.argv else dt.F16
shape = ([a for a in sys.argv[1:] if a != "q8"] or ["llama2_7b"])[0]
wk, _, s = synth.build(shape, dt.Q4_B32T1A, kv, max_ctx=2100)
rng = np.random.default_rng(1)
for n in (16, 48, 100, 150, 200, 256, 300, 400, 450, 500, 600, 1024, 2048):
    pr = rng.integers(3, s["vocab"], n).astype(np.int32)
    tok = wk.forward(pr, 0); wk.decode(tok, n, 4)
    best = 0.0
    for r3 in range(3):
        toks, ms = wk.decode(tok, n, 16); best = max(best, 16e3 / ms)
    print(json.dumps({"context": n, "kv": dt.name(kv), "shape": shape, "tok_s": round(best, 1)}), flush=True)
