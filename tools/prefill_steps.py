#!/usr/bin/env python3
"""A few prefills of one prompt length (for rocprofv3 --kernel-trace --stats): prefill_steps.py <shape> <tokens> <reps> [wdtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth
shape, T, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
wk, _, s = synth.build(shape, dt.Q4_B32T1A, dt.F16, max_ctx=max(2048, T + 16))
if os.environ.get("IFA_NO_SPLITK"):      # A / B: the large-tile kernel without its split-K form
    from inferflow_amd import _capi
    _capi.lib().ifa_gemm_big_tiles(1 | (1 << 12))
for opt in ("prefill_mid_max", "prefill_res_mid", "prefill_mid"):      # A / B of a route option: IFA_PREFILL_MID_MAX=4096 ...
    if os.environ.get("IFA_" + opt.upper()):
        wk.set_option(opt, int(os.environ["IFA_" + opt.upper()]))
rng = np.random.default_rng(3)
prompt = rng.integers(3, s["vocab"], T).astype(np.int32)
for rep in range(reps):
    wk.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wk.forward(prompt, 0)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("%s prefill %d tokens: %.3f ms = %.0f tok/s" % (shape, T, el * 1e3, T / el), flush=True)
