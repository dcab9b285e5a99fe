#!/usr/bin/env python3
"""Decode at a given context: prefill N tokens, 8 + 32 steps; prints the rate, the ids and a checksum of the last logits row (used to compare
two builds bit for bit: IFA_LIB=... python tools/ctx_prof.py 4096 [q8]) -- also the workload of the long-context kernel profiles."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
kv = dt.Q8_B32T2 if "q8" in sys.argv else dt.F16
n = int(sys.argv[1])
shape = os.environ.get("IFA_SHAPE", "llama2_7b")
wk, _, s = synth.build(shape, dt.Q4_B32T1A, kv, max_ctx=n + 100)
for o in (os.environ.get("IFA_OPTS") or "").split(","):      # IFA_OPTS="attn_nsplits=16,attn_fold_combine=0"
    if "=" in o: wk.set_option(o.split("=")[0], int(o.split("=")[1]))
pr = np.random.default_rng(1).integers(3, s["vocab"], n).astype(np.int32)
tok = wk.forward(pr, 0)
t8, _ = wk.decode(tok, n, 8)
toks, ms = wk.decode(int(t8[-1]), n + 8, 32)
lg = wk.read_buffer("logits")
print("ctx %d %s: %.1f tok/s  ids %s  logits crc %08x" % (n, dt.name(kv), 32e3 / ms, " ".join(str(int(t)) for t in toks[:8]), zlib.crc32(lg.tobytes())))
