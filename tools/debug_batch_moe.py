#!/usr/bin/env python3
"""Diagnosis aid (GPU box): the batched decode step of an MoE model at full width, option set by option set, every set run TWICE on a
fixed (teacher-forced) token sequence: is the step deterministic, and which option removes an outlier row?
    python tools/debug_batch_moe.py [shape] [layers] [queries] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from inferflow_amd import dtypes as dt, synth
from tests import gpu_util as g


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "mixtral_8x7b"
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    wk, _, s = synth.build(shape, dt.Q4_B32T1A, dt.F16, max_ctx=64, layers=layers)
    V = s["vocab"]
    wk.kv_slots(n)
    rng = np.random.default_rng(87)
    prompts = [rng.integers(3, V, 2 + i % 3).astype(np.int32) for i in range(n)]
    seq = rng.integers(3, V, (steps, n)).astype(np.int32)
    sets = [("default", {}), ("default again", {}), ("moe_overlap=0", {"moe_overlap": 0}), ("moe_singles=0", {"moe_singles": 0}), ("rows_mo=0", {"rows_mo": 0}),
            ("moe_device=0", {"moe_device": 0}), ("batch_fused=0", {"batch_fused": 0}), ("batch_fused=0 again", {"batch_fused": 0}),
            ("graph replay (no logits)", {"_graph": 1})]
    res = {}
    for name, opts in sets:
        for k, v in opts.items():
            if not k.startswith("_"):
                wk.set_option(k, v)
        wk.reset()
        pos = []
        for i, pr in enumerate(prompts):
            wk.select_kv(i); wk.forward(pr, 0); pos.append(len(pr))
        out = np.zeros((steps, n, V), np.float16)
        toks = np.zeros((steps, n), np.int32)
        lg = torch.empty((n, V), dtype=torch.float16, device="cuda")
        for st in range(steps):
            if opts.get("_graph"):
                toks[st] = wk.decode_batch(seq[st], pos, list(range(n)))
            else:
                toks[st] = wk.decode_batch(seq[st], pos, list(range(n)), lg)
                out[st] = g.host(lg)
            pos = [p + 1 for p in pos]
        res[name] = (out, toks)
        for k in opts:
            if not k.startswith("_"):
                wk.set_option(k, 1)
    base = res["batch_fused=0"][0].astype(np.float32)
    for name, (out, toks) in res.items():
        if name.startswith("graph"):
            agree = (toks == res["default"][1]).mean()
            print("%-26s greedy ids equal to the eager default: %.3f" % (name, agree))
            continue
        a = out.astype(np.float32)
        cos = (a * base).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(base, axis=-1) + 1e-30)
        bad = [(int(st), int(q), round(float(cos[st, q]), 4)) for st in range(steps) for q in range(n) if cos[st, q] < 0.999]
        print("%-26s min cos vs op-by-op rows %.5f, median %.6f; rows below 0.999: %s" % (name, cos.min(), np.median(cos), bad), flush=True)
    wk.close()


if __name__ == "__main__":
    main()
