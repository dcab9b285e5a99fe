#!/usr/bin/env python3
"""Prefill rate by prompt length (Llama-2-7B Q4): the four large-tile launches per layer from `prefill_big_min` + 1 tokens on
(option; 128 = round 4's rule) against the op-by-op layer below it.  IFA_PROMPT_LENS="48,64,96,128"  IFA_BIG_MINS="128,32" """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth

lens = [int(v) for v in os.environ.get("IFA_PROMPT_LENS", "40,48,64,96,128,192,256").split(",")]
mins = [int(v) for v in os.environ.get("IFA_BIG_MINS", "128,47").split(",")]
OPT = os.environ.get("IFA_AB_OPTION", "prefill_big_min")      # the option whose values IFA_BIG_MINS lists (e.g. prefill_chunk with 0,1)
wk, _, s = synth.build(os.environ.get("IFA_SHAPE", "llama2_7b"), dt.Q4_B32T1A, dt.F16, max_ctx=max(lens) + 8)
V = s["vocab"]
for T in lens:
    toks = np.random.default_rng(T).integers(3, V, T).astype(np.int32)
    out = {}
    for mn in mins:
        wk.set_option(OPT, mn)
        lg = torch.empty((T, V), dtype=torch.float16, device="cuda")
        wk.reset(); tok = wk.forward(toks, 0, lg)
        last = lg[-1].float().cpu().numpy()
        n = 10
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            wk.reset(); wk.forward(toks, 0)
        torch.cuda.synchronize()
        out[mn] = ((time.perf_counter() - t0) / n, int(tok), last)
    ref = out[mins[0]]
    print("T=%4d " % T + "  ".join("%s %3d: %.3f ms (%.0f tok/s)" % (OPT, mn, out[mn][0] * 1e3, T / out[mn][0]) for mn in mins)
          + "  same token: %s  max|dlogit| %.4f (std %.3f)" % (all(out[mn][1] == ref[1] for mn in mins),
                                                                max(float(np.abs(out[mn][2] - ref[2]).max()) for mn in mins), float(ref[2].std())), flush=True)
