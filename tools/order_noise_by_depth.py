#!/usr/bin/env python3
"""tools/order_noise.py by model depth: N-layer models (Llama-2-7B widths), the timed path against the order-exact path, teacher-forced
with the exact path's tokens -- how the half-ulp differences of one op's summation order (tests/test_gpu_fullsize_oracle.py:
<= 1 half ulp on < 0.5 % of an op's outputs) grow with the number of int8 re-quantisations between them and the logits.

    python tools/order_noise_by_depth.py [q4|q3h] [f16|q8] [steps]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth

wd = dt.Q3H_B64T1 if len(sys.argv) > 1 and sys.argv[1] == "q3h" else dt.Q4_B32T1A
kv = dt.Q8_B32T2 if len(sys.argv) > 2 and sys.argv[2] == "q8" else dt.F16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 48
print("%s weights, %s KV cache, %d teacher-forced steps per depth; max|dlogit| / std(logits) and relative RMS error of the logits" % (dt.name(wd), dt.name(kv), steps))
for N in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    we, _, s = synth.build("llama2_7b", wd, kv, max_ctx=steps + 8, layers=N)
    wt, _, _ = synth.build("llama2_7b", wd, kv, max_ctx=steps + 8, layers=N)
    we.set_option("exact_order", 1)
    cur = 11
    mads, rels, same = [], [], 0
    for i in range(steps):
        te, _ = we.decode(cur, i, 1)
        tt, _ = wt.decode(cur, i, 1)
        le = we.read_buffer("logits").view(np.float16).astype(np.float32)
        lt = wt.read_buffer("logits").view(np.float16).astype(np.float32)
        std = float(le.std())
        mads.append(float(np.abs(le - lt).max()) / std)
        rels.append(float(np.linalg.norm(le - lt) / np.linalg.norm(le - le.mean())))
        same += int(te[0]) == int(tt[0])
        cur = int(te[0])
    print("  %2d layers: max|d| / std median %.4f  max %.4f | relative RMS error median %.5f  max %.5f | bit-identical logits rows %d | same id %d of %d"
          % (N, np.median(mads), max(mads), np.median(rels), max(rels), sum(1 for m in mads if m == 0.0), same, steps), flush=True)
    we.close(); wt.close()
