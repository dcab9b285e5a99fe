#!/usr/bin/env python3
"""Per-phase timeline of the four rows-GEMM launches of a batched decode step (IFA_ROWS_TRACE=1 must be set: the launcher then
synchronises after every launch and prints the workgroup stamps).  python tools/trace_batch_rows.py [queries]"""
import os, sys
os.environ.setdefault("IFA_ROWS_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=256)
wk.kv_slots(n)
wk.set_option("graph", 0)
rng = np.random.default_rng(3)
cur = []
for i in range(n):
    wk.select_kv(i)
    cur.append(wk.forward(rng.integers(3, s["vocab"], 16).astype(np.int32), 0))
pos = [16] * n
for st in range(3):
    print("=== step", st, file=sys.stderr, flush=True)
    cur = [int(t) for t in wk.decode_batch(cur, pos, list(range(n)))]
    pos = [p + 1 for p in pos]
