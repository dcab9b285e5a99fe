#!/usr/bin/env python3
"""A few host-driven batched decode steps (for rocprofv3 --kernel-trace + tools/gap_from_trace.py):
    batch_steps.py <shape> <queries> <steps> [wdtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth
shape, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
wk, _, s = synth.build(shape, dt.Q4_B32T1A, dt.F16, max_ctx=256)
wk.kv_slots(n)
if os.environ.get("IFA_NO_GRAPH"): wk.set_option("graph", 0)      # eager launches (IFA_ROWS_TRACE=1 needs them)
rng = np.random.default_rng(3)
cur = []
for i in range(n):
    wk.select_kv(i)
    cur.append(int(wk.forward(rng.integers(3, s["vocab"], 16).astype(np.int32), 0)))
pos = [16] * n
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for st in range(steps):
        cur = [int(t) for t in wk.decode_batch(cur, pos, list(range(n)))]
        pos = [p + 1 for p in pos]
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("%s queries %d: %.3f ms per step" % (shape, n, el * 1e3 / steps), flush=True)
