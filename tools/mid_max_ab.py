#!/usr/bin/env python3
"""Prompts above prefill_mid_max (256): the large-tile GEMM (default) against the mid-size GEMM (prefill_mid_max raised), one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=1100)
for T in [int(v) for v in os.environ.get("IFA_PROMPT_LENS", "256,320,384,512,768,1024").split(",")]:
    toks = np.random.default_rng(T).integers(3, s["vocab"], T).astype(np.int32)
    res = []
    for mx in (256, 4096):
        wk.set_option("prefill_mid_max", mx)
        wk.reset(); tok = wk.forward(toks, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            wk.reset(); wk.forward(toks, 0)
        torch.cuda.synchronize()
        res.append("mid_max %d: %.3f ms (%.0f tok/s) tok %d" % (mx, (time.perf_counter() - t0) / 5 * 1e3, T / ((time.perf_counter() - t0) / 5), tok))
    print("T=%d  " % T + "  ".join(res), flush=True)
wk.set_option("prefill_mid_max", 256)
