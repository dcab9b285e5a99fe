#!/usr/bin/env python3
"""A / B of one engine option on one model: tokens and last-step logits of a greedy decode must be bit-identical between
the two settings; prints ms per token of each (several rounds, alternating) and the per-kernel HIP-event times.

    python tools/ab_option.py <option> [shape] [wdtype] [kv_dtype] [--steps N] [--prompt N] [--rounds N] [--values a,b]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from inferflow_amd import dtypes as dt, synth

DT = {"q4": dt.Q4_B32T1A, "q3h": dt.Q3H_B64T1, "q8": dt.Q8_B32T2, "f16": dt.F16, "q4b64": dt.Q4_B64T1, "q5": dt.Q5_B64T1, "q6": dt.Q6_B64T1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("option")
    ap.add_argument("shape", nargs="?", default="llama2_7b")
    ap.add_argument("wdtype", nargs="?", default="q4")
    ap.add_argument("kv_dtype", nargs="?", default="f16")
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--prompt", type=int, default=16)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--values", default="0,1")
    ap.add_argument("--max-ctx", type=int, default=512)
    ap.add_argument("--kernels", action="store_true")
    a = ap.parse_args()
    vals = [int(v) for v in a.values.split(",")]
    wk, _, s = synth.build(a.shape, DT[a.wdtype], DT[a.kv_dtype], max_ctx=a.max_ctx)
    prompt = (np.arange(a.prompt, dtype=np.int32) * 7 + 3) % s["vocab"]
    res = {}
    for rnd in range(a.rounds):
        for v in vals:
            wk.set_option(a.option, v)
            wk.reset()
            tok = wk.forward(prompt, 0)
            toks, ms = wk.decode(int(tok), a.prompt, a.steps)
            logits = wk.read_buffer("logits").view(np.uint16).copy()
            r = res.setdefault(v, {"ms": [], "toks": toks, "logits": logits})
            r["ms"].append(ms / a.steps)
            assert list(toks) == list(r["toks"]) and np.array_equal(logits, r["logits"]), "option %s=%d is not deterministic" % (a.option, v)
    base = res[vals[0]]
    for v in vals:
        r = res[v]
        same = list(r["toks"]) == list(base["toks"]) and np.array_equal(r["logits"], base["logits"])
        print("%s=%d: ms/token %s  best %.4f (%.1f tok/s)  tokens+logits identical to %s=%d: %s" % (
            a.option, v, " ".join("%.4f" % x for x in r["ms"]), min(r["ms"]), 1000.0 / min(r["ms"]), a.option, vals[0], same), flush=True)
        if not same:
            nd = int((r["logits"] != base["logits"]).sum())
            print("   MISMATCH: %d logits differ; tokens %s vs %s" % (nd, list(r["toks"])[:12], list(base["toks"])[:12]))
    if a.kernels:
        names = ["qkv", "attn", "wo", "ffn13", "w2", "lm_head"]
        for v in vals:
            wk.set_option(a.option, v)
            print("%s=%d kernels (us): %s" % (a.option, v, "  ".join("%s %.2f" % (n, wk.time_kernel(i, 64)) for i, n in enumerate(names))), flush=True)
    bad = [v for v in vals if not (list(res[v]["toks"]) == list(base["toks"]) and np.array_equal(res[v]["logits"], base["logits"]))]
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
