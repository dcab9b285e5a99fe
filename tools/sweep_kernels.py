#!/usr/bin/env python3
"""Sweep workgroups-per-CU for the fused decode kernels (GPU box only)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=256)
prompt = np.arange(3, 19, dtype=np.int32)
tok = wk.forward(prompt, 0)
names = ["qkv", "attn", "wo", "ffn13", "w2", "lm_head"]
opts = ["rpw_qkv", None, "rpw_wo", "rpw_ffn", "rpw_w2", "rpw_lm"]
for per_cu in (1, 2, 3, 4):
    row = {"wgs_per_cu": per_cu}
    for i, nm in enumerate(names):
        if opts[i]:
            wk.set_option(opts[i], per_cu)
        row[nm] = round(wk.time_kernel(i, 200), 2)
    print(json.dumps(row), flush=True)
# whole-token timing at the best-looking settings is left to bench.py
