#!/usr/bin/env python3
"""Phase stamps of the fused QKV + attention launch (option trace): per workgroup start / rows published, per head the
attention tail's stamps, all relative to the earliest workgroup start of the launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inferflow_amd import dtypes as dt, synth
kv = dt.Q8_B32T2 if len(sys.argv) > 1 and sys.argv[1] == "q8" else dt.F16
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, kv, max_ctx=256)
NP = int(os.environ.get("IFA_TRACE_CTX", "16"))       # prompt length: the timing launches then sit at NP + 48 keys
tok = wk.forward((np.arange(NP, dtype=np.int32) % 1000) + 3, 0)
toks, ms = wk.decode(int(tok), NP, 48 if NP == 16 else 8)          # (16: positions up to 64, the prefetch bucket of the timing launches)
wk.set_option("trace", 1)
H = s["heads"]
for which, nm in [(7, "qkv+attn fused"), (0, "qkv"), (1, "attn"), (2, "wo"), (3, "ffn13")]:
    us = wk.time_kernel(which, 33)
    if False:
        tr = wk.read_buffer("trace").view(np.int64).reshape(2048, 8)[:256]
        t0 = tr[:, 0].min()
        fr, ld = tr[tr[:, 7] == 1], tr[tr[:, 7] == 0]
        def col(a, i):
            return "%.2f (%.2f)" % (float(np.median(a[:, i] - t0)) * 0.01, (a[:, i].max() - t0) * 0.01)
        print("%s event_us %.2f | us after the launch's first instruction, median (max)" % (nm, us))
        print("   %d front workgroups : Wo rows in memory %s | all Wo rows seen %s | image quantised %s | slice + flag out %s" % (len(fr), col(fr, 1), col(fr, 2), col(fr, 3), col(fr, 4)))
        print("   %d loader workgroups: rows requested %s | image flags seen %s | image in LDS %s | rows stored %s" % (len(ld), col(ld, 1), col(ld, 2), col(ld, 3), col(ld, 4)))
        continue
    if which != 7:
        print("%s event_us %.2f" % (nm, us)); continue
    tr = wk.read_buffer("trace").view(np.int64).reshape(2048, 8)
    wg = tr[H:H + 256, :2]
    t0 = wg[:, 0].min()
    att = tr[:H]
    lab = ["prefetch_issued", "staged", "rope_kv_kt", "scores", "max", "probs", "pv", "end"]
    print("%s event_us %.2f | workgroup start skew %.2f, rows published: median %.2f max %.2f" % (
        nm, us, (wg[:, 0].max() - t0) * 0.01, float(np.median(wg[:, 1] - t0)) * 0.01, (wg[:, 1].max() - t0) * 0.01))
    print("   attention tail (us after the launch's first instruction), median over heads: " +
          "  ".join("%s %.2f" % (lab[i], float(np.median(att[:, i] - t0)) * 0.01) for i in range(8)))
    print("   last head ends at %.2f" % ((att[:, 7].max() - t0) * 0.01))
    wo = tr[H:H + 256]
    wo = wo[wo[:, 4] > 0]
    if len(wo):
        print("   Wo tail of the %d other workgroups: flags seen median %.2f max %.2f | image gathered median %.2f max %.2f | rows stored median %.2f max %.2f" % (
            len(wo), float(np.median(wo[:, 2] - t0)) * 0.01, (wo[:, 2].max() - t0) * 0.01, float(np.median(wo[:, 3] - t0)) * 0.01, (wo[:, 3].max() - t0) * 0.01,
            float(np.median(wo[:, 4] - t0)) * 0.01, (wo[:, 4].max() - t0) * 0.01))
