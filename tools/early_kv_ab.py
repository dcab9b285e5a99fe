#!/usr/bin/env python3
"""Decode rate by context up to the split threshold, and the CRC of the logits: the head's K / V rows requested behind the attention
waves' weight rows (IFA_QA_EARLY_KV=1, default) against requested after the workgroup's last row (IFA_LIB=lib_variants/latekv/...).
    early_kv_ab.py [q8]"""
import json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
kv = dt.Q8_B32T2 if "q8" in sys.argv else dt.F16
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, kv, max_ctx=512)
rng = np.random.default_rng(1)
for n in (16, 48, 100, 150, 200, 250, 300):
    pr = rng.integers(3, s["vocab"], n).astype(np.int32)
    tok = wk.forward(pr, 0)
    wk.decode(tok, n, 4)
    best = 0.0
    for rep in range(3):
        toks, ms = wk.decode(tok, n, 16)
        best = max(best, 16e3 / ms)
    crc = zlib.crc32(np.asarray(toks, dtype=np.int32).tobytes())
    print(json.dumps({"context": n, "kv": dt.name(kv), "decode_tok_s": round(best, 1), "ids_crc": crc}), flush=True)
