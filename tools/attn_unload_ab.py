#!/usr/bin/env python3
"""Option attn_unload (the heads' workgroups of the fused QKV + attention launch take no weight rows in the 256-row bucket) against
the default, one process, alternating: rate of a 16-step decode call and the CRC of its greedy ids by prompt length.
    attn_unload_ab.py [q8] [shape]"""
import json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
kv = dt.Q8_B32T2 if "q8" in sys.argv else dt.F16
shape = [a for a in sys.argv[1:] if a not in ("q8",)]
shape = shape[0] if shape else "llama2_7b"
wk, _, s = synth.build(shape, dt.Q4_B32T1A, kv, max_ctx=512)
rng = np.random.default_rng(1)
for n in [int(v) for v in os.environ.get('IFA_AB_CTX', '100,130,150,200,250,300').split(',')]:
    pr = rng.integers(3, s["vocab"], n).astype(np.int32)
    row = {"context": n, "kv": dt.name(kv), "shape": shape}
    for rep in range(2):
        for ul in [int(v) for v in os.environ.get('IFA_AB_UL', '0,1').split(',')]:
            wk.set_option("attn_unload", ul)
            tok = wk.forward(pr, 0)
            wk.decode(tok, n, 4)
            best = 0.0
            for r3 in range(3):
                toks, ms = wk.decode(tok, n, 16)
                best = max(best, 16e3 / ms)
            row["tok_s_ul%d_rep%d" % (ul, rep)] = round(best, 1)
            row["crc_ul%d" % ul] = zlib.crc32(np.asarray(toks, dtype=np.int32).tobytes())
    row["ids_equal"] = len({v for k, v in row.items() if k.startswith("crc_")}) == 1
    print(json.dumps(row), flush=True)
