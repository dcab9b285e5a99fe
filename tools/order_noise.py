#!/usr/bin/env python3
"""The timed decode path against the order-exact path ON THE GPU (option exact_order: bit-identical to the CPU oracle,
tests/test_gpu_fullsize_oracle.py), teacher-forced with the exact path's tokens, over many more steps and longer contexts than the CPU
oracle affords: what the wave64 summation order of the timed kernels costs against the reference order, per step.

    python tools/order_noise.py [q4|q3h] [f16|q8] [steps] [prompt tokens]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth

wd = dt.Q3H_B64T1 if len(sys.argv) > 1 and sys.argv[1] == "q3h" else dt.Q4_B32T1A
kv = dt.Q8_B32T2 if len(sys.argv) > 2 and sys.argv[2] == "q8" else dt.F16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 256
n_prompt = int(sys.argv[4]) if len(sys.argv) > 4 else 16
ctx = n_prompt + steps + 8
we, _, s = synth.build("llama2_7b", wd, kv, max_ctx=ctx)
wt, _, _ = synth.build("llama2_7b", wd, kv, max_ctx=ctx)
we.set_option("exact_order", 1)
prompt = np.random.default_rng(77).integers(3, s["vocab"], n_prompt).astype(np.int32)
cur = None
rows = []
for i in range(n_prompt + steps):
    tok_in = int(prompt[i]) if i < n_prompt else cur
    te, _ = we.decode(tok_in, i, 1)
    tt, _ = wt.decode(tok_in, i, 1)                # the timed path: graph replay of the four-launch layer, every step a T = 1 step
    le = we.read_buffer("logits").view(np.float16).astype(np.float32)
    lt = wt.read_buffer("logits").view(np.float16).astype(np.float32)
    std = float(le.std())
    cos = float((le * lt).sum() / (np.linalg.norm(le) * np.linalg.norm(lt) + 1e-30))
    mad = float(np.abs(le - lt).max()) / std
    top2 = np.partition(le, -2)[-2:]
    rows.append((i, cos, mad, int(te[0]) == int(tt[0]), float(top2[1] - top2[0]) / std))
    cur = int(te[0])
r = np.array([(c, m, a, g) for _, c, m, a, g in rows], np.float64)
print("%s weights, %s KV cache, %d teacher-forced single-token steps (contexts 1..%d), 32 layers; reference = the order-exact path" % (dt.name(wd), dt.name(kv), len(rows), len(rows)))
print("  max|dlogit| / std(logits): median %.4f  90%% %.4f  99%% %.4f  max %.4f" % tuple(np.quantile(r[:, 1], q) for q in (0.5, 0.9, 0.99, 1.0)))
print("  cosine: median %.6f  min %.6f" % (np.median(r[:, 0]), r[:, 0].min()))
dis = [(i, g) for i, _, _, a, g in rows if not a]
print("  greedy id equal on %d of %d steps; the %d others have top-2 gaps of %s x std (the largest: %.3f)" % (int(r[:, 2].sum()), len(rows), len(dis),
      ", ".join("%.3f" % g for _, g in dis[:12]) + (" ..." if len(dis) > 12 else ""), max([g for _, g in dis], default=0.0)))
for lo in range(0, len(rows), max(1, len(rows) // 8)):
    seg = r[lo:lo + max(1, len(rows) // 8)]
    print("  steps %4d..%4d: median |dlogit| / std %.4f  max %.4f" % (lo, lo + len(seg) - 1, np.median(seg[:, 1]), seg[:, 1].max()))
