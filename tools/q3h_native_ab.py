#!/usr/bin/env python3
"""A / B of the native 32-byte Q3H_B64T1 stream (option q3h_native, csrc/ifa_decode_formats.h WRowQ3HN) against the 36-byte nibble-pair
stream on configs[2] (Llama-2-7B widths, Q3H_B64T1 weights, Q8_B32T2 KV cache): HIP-event averages of the Wo / W1 | W3 / W2 launches
(rotating over the 32 layers) and decode tokens/s, alternating on ONE box.

    python tools/q3h_native_ab.py [rounds]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wk, _, s = synth.build("llama2_7b", dt.Q3H_B64T1, dt.Q8_B32T2, max_ctx=256)
prompt = np.arange(3, 19, dtype=np.int32)
ref = None
for r in range(rounds):
    for native in (0, 1):
        wk.set_option("q3h_native", native)
        wk.reset()
        tok = wk.forward(prompt, 0)
        wk.decode(int(tok), 16, 8)                      # capture + warm
        wk.reset(); tok = wk.forward(prompt, 0)
        toks, ms = wk.decode(int(tok), 16, 128)
        if ref is None:
            ref = list(toks)
        k = {nm: wk.time_kernel(which, 96) for which, nm in ((2, "wo"), (3, "w13"), (4, "w2"))}
        bytes_w13 = 2 * 11008 * 64 * (32 if native else 36)
        print("round %d q3h_native=%d: %.1f tok/s (%.4f ms/token) | wo %.2f us  w1|w3 %.2f us (%.0f GB/s streamed, %.0f GB/s algorithmic)  w2 %.2f us | tokens %s"
              % (r, native, 128 / ms * 1e3, ms / 128, k["wo"], k["w13"], bytes_w13 / k["w13"] / 1e3, 2 * 11008 * 64 * 32 / k["w13"] / 1e3, k["w2"],
                 "same" if list(toks) == ref else "DIFFER"), flush=True)
wk.set_option("q3h_native", 0)
