#!/usr/bin/env python3
"""Prefill rate of short prompts (2..16 tokens): the fused four-GEMM layer against the op-by-op layer (batch_fused = 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth

wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=256)
for T in (2, 4, 8, 12, 16):
    toks = np.arange(3, 3 + T, dtype=np.int32)
    out = {}
    for fused in (1, 0):
        wk.set_option("batch_fused", fused)
        first = wk.forward(toks, 0)
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            tok = wk.forward(toks, 0)
        dtm = (time.perf_counter() - t0) / n
        out[fused] = (dtm, tok)
    print("T=%2d fused %.3f ms (%.0f tok/s)  op-by-op %.3f ms (%.0f tok/s)  same first token: %s" % (
        T, out[1][0] * 1e3, T / out[1][0], out[0][0] * 1e3, T / out[0][0], out[1][1] == out[0][1]), flush=True)
