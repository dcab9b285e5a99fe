#!/usr/bin/env python3
"""One workgroup per head (tail of the fused launch) against keys split over workgroups, by context: option attn_split_ctx 320 / 1000.
    split_threshold_ab.py [q8] [shape]"""
import json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
kv = dt.Q8_B32T2 if "q8" in sys.argv else dt.F16
shape = ([a for a in sys.argv[1:] if a != "q8"] or ["llama2_7b"])[0]
wk, _, s = synth.build(shape, dt.Q4_B32T1A, kv, max_ctx=1024)
rng = np.random.default_rng(1)
for n in (300, 400, 450, 500, 560, 620, 700, 800):
    pr = rng.integers(3, s["vocab"], n).astype(np.int32)
    row = {"context": n, "kv": dt.name(kv), "shape": shape}
    for thr in (320, 1000):
        wk.set_option("attn_split_ctx", thr)
        tok = wk.forward(pr, 0); wk.decode(tok, n, 4)
        best = 0.0
        for r3 in range(3):
            toks, ms = wk.decode(tok, n, 16); best = max(best, 16e3 / ms)
        row["split_from_%d" % thr] = round(best, 1)
    print(json.dumps(row), flush=True)
