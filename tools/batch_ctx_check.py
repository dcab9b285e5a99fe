#!/usr/bin/env python3
"""Batched decode at a given context: n queries with prompts of CTX tokens, a few steps through ifa_model_decode_batch; prints the step time,
the ids and a checksum of the logits rows (two builds compared bit for bit: IFA_LIB=... python tools/batch_ctx_check.py 4 700 [q8])."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth
n, ctx = int(sys.argv[1]), int(sys.argv[2])
kv = dt.Q8_B32T2 if "q8" in sys.argv else dt.F16
wk, _, s = synth.build(os.environ.get("IFA_SHAPE", "llama2_7b"), dt.Q4_B32T1A, kv, max_ctx=ctx + 64)
wk.kv_slots(n)
rng = np.random.default_rng(5)
cur = []
for i in range(n):
    wk.select_kv(i)
    cur.append(int(wk.forward(rng.integers(3, s["vocab"], ctx - 7 * i).astype(np.int32), 0)))      # ragged contexts
pos = [ctx - 7 * i for i in range(n)]
lg = torch.empty((n, s["vocab"]), dtype=torch.float16, device="cuda")
crc = 0
for st in range(4):
    cur = [int(t) for t in wk.decode_batch(cur, pos, list(range(n)), lg)]
    crc = zlib.crc32(lg.cpu().numpy().tobytes(), crc)
    pos = [p + 1 for p in pos]
torch.cuda.synchronize(); t0 = time.perf_counter()
for st in range(16):
    cur = [int(t) for t in wk.decode_batch(cur, pos, list(range(n)))]
    pos = [p + 1 for p in pos]
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 16
print("batch %d ctx %d %s: %.3f ms per step  ids %s  logits crc %08x" % (n, ctx, dt.name(kv), ms, " ".join(str(t) for t in cur[:4]), crc))
