import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth, _capi
L = _capi.lib()
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=600)
for T in (64, 96, 128, 192, 256, 512):
    toks = np.random.default_rng(T).integers(3, s["vocab"], T).astype(np.int32)
    res = []
    for m in (64, 0, 128, 256):
        L.ifa_attention_two_pass_min(m)
        wk.reset(); tok = wk.forward(toks, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6):
            wk.reset(); wk.forward(toks, 0)
        torch.cuda.synchronize()
        res.append("min %d: %.3f ms tok %d" % (m, (time.perf_counter() - t0) / 6 * 1e3, tok))
    print("T=%d  " % T + "  ".join(res), flush=True)
L.ifa_attention_two_pass_min(64)
