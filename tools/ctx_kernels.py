#!/usr/bin/env python3
"""Where the decode step's time goes as the context grows (below the split threshold): per-launch HIP-event averages
(ifa_model_time_kernel) next to the rate of a 16-step decode call, by prompt length.    ctx_kernels.py [q8] [max_ctx]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
kv = dt.Q8_B32T2 if "q8" in sys.argv else dt.F16
mc = [int(a) for a in sys.argv[1:] if a.isdigit()]
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, kv, max_ctx=mc[0] if mc else 512)
if os.environ.get("IFA_CTXK_UNLOAD"): wk.set_option("attn_unload", int(os.environ["IFA_CTXK_UNLOAD"]))
rng = np.random.default_rng(1)
for n in (16, 100, 200, 250, 300):
    pr = rng.integers(3, s["vocab"], n).astype(np.int32)
    tok = wk.forward(pr, 0)
    wk.decode(tok, n, 4)
    best = 1e9
    for rep in range(3):
        toks, ms = wk.decode(tok, n, 16)
        best = min(best, ms / 16)
    k = {nm: round(wk.time_kernel(w, 50), 2) for w, nm in [(7, "qkv_attn"), (2, "wo"), (3, "ffn13"), (4, "w2"), (5, "lm_head")]}
    layer = k["qkv_attn"] + k["wo"] + k["ffn13"] + k["w2"]
    print(json.dumps({"context": n, "kv": dt.name(kv), "ms_per_step": round(best, 4), "tok_s": round(1e3 / best, 1), "kernels_us": k,
                      "layers_x32_plus_head_ms": round((32 * layer + k["lm_head"]) * 1e-3, 4)}), flush=True)
