#!/usr/bin/env python3
"""Prefill with / without the stream-K schedule of the 256 x 256 tiles (csrc/ifa_gemm.hip, k_gemm_big<.., KS = 0>; switch: bit 13 of
ifa_gemm_big_tiles: opt-in), alternating on one box:   stream_k_ab.py [tokens ...]   (default 1024 2048)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth, _capi
L = _capi.lib()
lens = [int(v) for v in sys.argv[1:]] or [1024, 2048]
wk, _, s = synth.build(os.environ.get("IFA_SHAPE", "llama2_7b"), dt.Q4_B32T1A, dt.F16, max_ctx=max(lens) + 16)
V = s["vocab"]
for T in lens:
    toks = np.random.default_rng(T).integers(3, V, T).astype(np.int32)
    res = {}
    for rnd in range(3):
        for name, mode in (("stream_k", 1 | (1 << 13)), ("off", 1)):
            L.ifa_gemm_big_tiles(mode)
            lg = torch.empty((T, V), dtype=torch.float16, device="cuda")
            wk.reset(); tok = wk.forward(toks, 0, lg)
            last = lg[-1].float().cpu().numpy()
            n = 6
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                wk.reset(); wk.forward(toks, 0)
            torch.cuda.synchronize()
            res.setdefault(name, []).append(((time.perf_counter() - t0) / n, int(tok), last))
    L.ifa_gemm_big_tiles(1)
    a, b = res["stream_k"], res["off"]
    print("T=%d  stream-K %s ms  off %s ms  (%.0f vs %.0f tok/s)  same token %s  max|dlogit| %.4f (std %.3f)" % (
        T, " / ".join("%.3f" % (r[0] * 1e3) for r in a), " / ".join("%.3f" % (r[0] * 1e3) for r in b),
        T / min(r[0] for r in a), T / min(r[0] for r in b), a[0][1] == b[0][1], float(np.abs(a[0][2] - b[0][2]).max()), float(b[0][2].std())), flush=True)
