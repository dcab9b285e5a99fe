#!/usr/bin/env python3
"""Phase stamps of the decode attention kernel (thread 0 of every head's workgroup), F16 and Q8 KV cache side by side.
    python tools/trace_attn.py [context]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 100
alab = ["start", "qkv_staged", "rope_kv", "scores", "max", "probs", "pv", "end"]
for name, kvd in (("f16", dt.F16), ("q8", dt.Q8_B32T2)):
    wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, kvd, max_ctx=256)
    tok = wk.forward((np.arange(ctx, dtype=np.int32) * 7 + 3) % s["vocab"], 0)
    wk.decode(tok, ctx, 4)
    wk.set_option("trace", 1)
    us = wk.time_kernel(1, 37)
    tr = wk.read_buffer("trace").view(np.int64).reshape(2048, 8)[:s["heads"]]
    rel = (tr - tr[:, :1]) * 0.01
    print("kv %s: event %.2f us | " % (name, us) + " ".join("%s=%.2f" % (alab[i], float(np.median(rel[:, i]))) for i in range(1, 8)), flush=True)
    wk.close()
