#!/usr/bin/env python3
"""Prompts of 33..64 tokens: two passes of the rows GEMM (prefill_chunk) against the mid-size GEMM route (prefill_big_min lowered), one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=128)
for T in (33, 34, 40, 47, 48, 56, 64):
    toks = np.random.default_rng(T).integers(3, s["vocab"], T).astype(np.int32)
    res = []
    for name, opts in (("default", {}), ("mid from 33", {"prefill_chunk": 0, "prefill_big_min": 32})):
        wk.set_option("prefill_chunk", 1); wk.set_option("prefill_big_min", 47)
        for k, v in opts.items(): wk.set_option(k, v)
        wk.reset(); tok = wk.forward(toks, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6):
            wk.reset(); wk.forward(toks, 0)
        torch.cuda.synchronize()
        res.append("%s: %.3f ms tok %d" % (name, (time.perf_counter() - t0) / 6 * 1e3, tok))
    print("T=%d  " % T + "  ".join(res), flush=True)
