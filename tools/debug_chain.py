#!/usr/bin/env python3
"""Debug aid for the chained FFN launch: one-layer model, one decode step per call; the FFN buffers of fuse_ffn = 1 / 2 against 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=256, layers=L)
prompt = (np.arange(16, dtype=np.int32) * 7 + 3) % s["vocab"]
def run(mode, steps=1):
    wk.set_option("fuse_ffn", mode)
    wk.reset()
    tok = wk.forward(prompt, 0)
    toks, _ = wk.decode(int(tok), len(prompt), steps)
    out = {n: wk.read_buffer(n).view(np.uint16).copy() for n in ("a", "t1", "x", "x2", "logits")}
    out["toks"] = list(toks)
    return out
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ref = run(0, STEPS)
for rep in range(3):
    for mode in (1, 2, 0):
        g = run(mode, STEPS)
        print("rep %d mode %d: " % (rep, mode) + "  ".join("%s %d" % (n, int((g[n] != ref[n]).sum())) for n in ("a", "t1", "x", "x2", "logits")) + "  toks %s" % (g["toks"] == ref["toks"]), flush=True)
        for n in ("t1", "x2", "x"):
            d = np.nonzero(g[n] != ref[n])[0]
            if len(d):
                print("    %s first diffs at %s ... got %s ref %s" % (n, d[:8], g[n][d[:4]], ref[n][d[:4]]))
