#!/usr/bin/env python3
"""Prefill time by prompt length with the large-tile GEMM's tile shape forced (measurement: ifa_gemm_big_tiles bits 8-10):
0 = the launcher's own choice, 3 = 128 x 128, 4 = 64 x 64, 5 = 128 x 64, 6 = 64 x 128 (tokens x weight rows).
IFA_PROMPT_LENS="64,128,256"  IFA_TILES="0,4,5,6" """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth, _capi

lens = [int(v) for v in os.environ.get("IFA_PROMPT_LENS", "48,64,128,256,512").split(",")]
tiles = [int(v) for v in os.environ.get("IFA_TILES", "0,4,5,6").split(",")]
wk, _, s = synth.build(os.environ.get("IFA_SHAPE", "llama2_7b"), dt.Q4_B32T1A, dt.F16, max_ctx=max(lens) + 8)
V = s["vocab"]
L = _capi.lib()
for T in lens:
    toks = np.random.default_rng(T).integers(3, V, T).astype(np.int32)
    out = {}
    for f in tiles:
        L.ifa_gemm_big_tiles(1 | (f << 8))
        lg = torch.empty((T, V), dtype=torch.float16, device="cuda")
        wk.reset(); tok = wk.forward(toks, 0, lg)
        last = lg[-1].float().cpu().numpy()
        n = 8
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            wk.reset(); wk.forward(toks, 0)
        torch.cuda.synchronize()
        out[f] = ((time.perf_counter() - t0) / n, int(tok), last)
    L.ifa_gemm_big_tiles(1)
    ref = out[tiles[0]]
    print("T=%4d " % T + "  ".join("tiles %d: %.3f ms (%.0f tok/s)" % (f, out[f][0] * 1e3, T / out[f][0]) for f in tiles)
          + "  same token: %s  max|dlogit| %.4f (std %.3f)" % (all(out[f][1] == ref[1] for f in tiles),
                                                                max(float(np.abs(out[f][2] - ref[2]).max()) for f in tiles), float(ref[2].std())), flush=True)
