#!/usr/bin/env python3
"""Prefill rate of long prompts: four large-tile GEMM launches per layer (prefill_big = 1, in-tree kernel) against the
op-by-op layer (prefill_big = 0: the smaller-tile kernels serve every T)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from inferflow_amd import dtypes as dt, synth

wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=4096)
V = s["vocab"]
rng = np.random.default_rng(3)
LENS = [int(v) for v in os.environ.get("IFA_PREFILL_LENS", "256,512,1024,2048,4096").split(",")]
MODES = [int(v) for v in os.environ.get("IFA_PREFILL_MODES", "1,0").split(",")]
for T in LENS:
    toks = rng.integers(3, V, T).astype(np.int32)
    out = {}
    for big in MODES:
        wk.set_option("prefill_big", big)
        lg = torch.empty((T, V), dtype=torch.float16, device="cuda") if T <= 512 else None
        first = wk.forward(toks, 0, lg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            tok = wk.forward(toks, 0)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / n
        out[big] = (dtm, tok, lg.float().cpu().numpy() if lg is not None else None)
    extra = ""
    if len(MODES) < 2:
        print("T=%4d prefill_big=%d %.2f ms (%.0f tok/s)" % (T, MODES[0], out[MODES[0]][0] * 1e3, T / out[MODES[0]][0]), flush=True)
        continue
    if out[1][2] is not None:
        a, b = out[1][2], out[0][2]
        extra = " logits cos %.6f max|d| %.4f" % (float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b))), float(np.abs(a - b).max()))
    print("T=%4d own large-tile %.2f ms (%.0f tok/s)  op-by-op %.2f ms (%.0f tok/s)  same token: %s%s" % (
        T, out[1][0] * 1e3, T / out[1][0], out[0][0] * 1e3, T / out[0][0], out[1][1] == out[0][1], extra), flush=True)
