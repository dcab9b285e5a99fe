#!/usr/bin/env python3
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=256)
tok = wk.forward(np.arange(3, 19, dtype=np.int32), 0)
wk.set_option("trace", 1)
labels = ["start", "loads_issued", "prologue_done", "xregs", "x_in_lds", "partials", "serial", "end"]
for mode, stride in [(0, 0)]:
    wk.set_option("bench_mode", mode)
    if stride:
        wk.set_option("touch_stride", stride)
    for which, nm in [(2, "wo"), (0, "qkv"), (3, "ffn13"), (4, "w2")]:
        us = wk.time_kernel(which, 37)
        tr = wk.read_buffer("trace").view(np.int64).reshape(2048, 8)
        tr = tr[tr[:, 0] > 0]
        t0 = tr[:, 0].min()
        rel = (tr - t0) * 0.01
        rel[tr == 0] = np.nan
        med = [float(np.nanmedian(rel[:, i])) if not np.all(np.isnan(rel[:, i])) else -1 for i in range(8)]
        print("mode", mode, "stride", stride, nm, "event_us %.1f" % us, "kernel_span %.2f" % float(np.nanmax(rel[:, 7])),
              "med:", " ".join("%s=%.1f" % (labels[i][:8], med[i]) for i in (1, 4, 5, 6, 2, 3, 7)), flush=True)

    # attention (one workgroup per head): stamps of thread 0
    alab = ["start", "qkv_staged", "rope_kv", "scores", "max", "probs", "pv", "end"]
    us = wk.time_kernel(1, 37)
    tr = wk.read_buffer("trace").view(np.int64).reshape(2048, 8)[:s["heads"]]
    rel = (tr - tr[:, :1]) * 0.01
    print("attn event_us %.1f" % us, "med:", " ".join("%s=%.2f" % (alab[i], float(np.median(rel[:, i]))) for i in range(1, 8)),
          "skew of starts %.2f" % ((tr[:, 0].max() - tr[:, 0].min()) * 0.01), flush=True)
