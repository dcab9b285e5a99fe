#!/usr/bin/env python3
"""Prompts above prefill_mid_max: wo / w2 through k_gemm_mid (option prefill_res_mid = the longest prompt that does so; here 1 << 20) against
all four products through the large-tile kernel (0), one box, alternating."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=4200)
for T in [int(v) for v in os.environ.get("IFA_PROMPT_LENS", "1024,1536,2048,4096").split(",")]:
    toks = np.random.default_rng(T).integers(3, s["vocab"], T).astype(np.int32)
    res, last = [], {}
    for rnd in range(2):
        for opt in (0, 1):
            wk.set_option("prefill_res_mid", (1 << 20) if opt else 0)
            lg = torch.empty((T, s["vocab"]), dtype=torch.float16, device="cuda")
            wk.reset(); tok = wk.forward(toks, 0, lg)
            last[opt] = (int(tok), lg[-1].float().cpu().numpy())
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4):
                wk.reset(); wk.forward(toks, 0)
            torch.cuda.synchronize()
            dtm = (time.perf_counter() - t0) / 4
            res.append("res_mid %d: %.3f ms (%.0f tok/s)" % (opt, dtm * 1e3, T / dtm))
    print("T=%d  " % T + "  ".join(res) + "  same token: %s  max|dlogit| %.4f (std %.3f)" % (last[0][0] == last[1][0], float(np.abs(last[0][1] - last[1][1]).max()), float(last[0][1].std())), flush=True)
wk.set_option("prefill_res_mid", 2048)
