#!/usr/bin/env python3
"""Aggregate decode rate with dynamic batching (Llama-2-7B Q4): n queries advance one token per step."""
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth
shape = sys.argv[1] if len(sys.argv) > 1 else "llama2_7b"
CTX = int(os.environ.get("IFA_BATCH_CTX", "16"))      # prompt length of every query (the context the steps start at)
wk, _, s = synth.build(shape, dt.Q4_B32T1A, dt.Q8_B32T2 if os.environ.get("IFA_BATCH_KV") == "q8" else dt.F16, max_ctx=max(256, CTX + 80))
NMAX = int(os.environ.get("IFA_BATCH_NMAX", "32"))
wk.kv_slots(NMAX)
if "graph" in sys.argv: wk.set_option("batch_graph", 1)
if os.environ.get("IFA_BATCH_FUSED") == "0": wk.set_option("batch_fused", 0)      # the op-by-op rows for comparison
for kv in filter(None, os.environ.get("IFA_BATCH_OPTS", "").split(",")):      # e.g. IFA_BATCH_OPTS=rows_norm32=0,rows_kparts=0 (A / B)
    wk.set_option(kv.split("=")[0], int(kv.split("=")[1]))
rng = np.random.default_rng(3)
first = []
for i in range(NMAX):
    wk.select_kv(i)
    first.append(wk.forward(rng.integers(3, s["vocab"], CTX).astype(np.int32), 0))
for n in [int(v) for v in os.environ.get("IFA_BATCH_SIZES", "1,2,4,8,16,32").split(",")]:
    cur, pos = list(first[:n]), [CTX] * n
    steps = 24
    for w in range(2):
        gc.collect(); gc.disable()          # host-driven steps: no 35-60 ms cyclic-collector pause inside the timed pass
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for st in range(steps):
            if n == 1:
                wk.select_kv(0)
                cur = [int(wk.decode(cur[0], pos[0], 1)[0][0])]
            else:
                cur = [int(t) for t in wk.decode_batch(cur, pos, list(range(n)))]
            pos = [p + 1 for p in pos]
        torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
        gc.enable()
    print(json.dumps({"shape": shape, "queries": n, "context": CTX, "ms_per_step": dt_s * 1e3 / steps, "aggregate_tok_s": n * steps / dt_s}), flush=True)
