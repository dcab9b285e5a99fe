#!/usr/bin/env python3
"""Phase stamps of the chained FFN launch (csrc/ifa_decode_chain.h, option fuse_ffn = 1 / 2) next to the HIP-event times of the
launches it replaces, all relative to the earliest workgroup start of the traced launch.

    python tools/trace_chain.py [q4|q3h] [f16|q8]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth

wd = dt.Q3H_B64T1 if len(sys.argv) > 1 and sys.argv[1] == "q3h" else dt.Q4_B32T1A
kv = dt.Q8_B32T2 if len(sys.argv) > 2 and sys.argv[2] == "q8" else dt.F16
wk, _, s = synth.build("llama2_7b", wd, kv, max_ctx=256)
tok = wk.forward(np.arange(3, 19, dtype=np.int32), 0)
toks, ms = wk.decode(int(tok), 16, 48)
base = {}
for which, nm in [(2, "wo"), (3, "ffn13"), (4, "w2")]:
    base[nm] = wk.time_kernel(which, 64)
print("separate launches (HIP events, us): " + "  ".join("%s %.2f" % kv_ for kv_ in base.items()))
LAB = ["start", "first requests out", "Wo rows published", "FFN input gathered", "FFN image in registers", "gated rows published",
       "flags of the gated rows seen", "gated rows gathered", "W2 image in LDS", "workgroup flag (gated rows) raised",
       "loader: FFN rows requested", "loader: FFN image in registers", "loader: gated rows published", "loader: W2 rows requested",
       "loader: W2 image seen", "loader: end"]
for mode, late in ((1, 0), (2, 0), (1, 1)):
    wk.set_option("fuse_ffn", mode)
    wk.set_option("chain_late_w2", late)
    wk.set_option("trace", 0)
    us = wk.time_kernel(9, 64)
    wk.set_option("trace", 1)
    us_t = wk.time_kernel(9, 33)
    tr = wk.read_buffer("trace").view(np.int64).reshape(-1, 16)[:256]
    t0 = tr[:, 0].min()
    ref = base["ffn13"] + base["w2"] + (base["wo"] if mode == 2 else 0.0)
    print("fuse_ffn=%d chain_late_w2=%d: %.2f us per launch (traced run %.2f) against %.2f for the launches it replaces" % (mode, late, us, us_t, ref))
    print("   workgroup start skew %.2f us" % ((tr[:, 0].max() - t0) * 0.01))
    for i in list(range(1, 16)):
        col = tr[:, i] - t0
        print("   %-32s median %6.2f  min %6.2f  max %6.2f" % (LAB[i], float(np.median(col)) * 0.01, col.min() * 0.01, col.max() * 0.01))
wk.set_option("trace", 0)
wk.set_option("chain_late_w2", 0)
