#!/usr/bin/env python3
"""Golden rows of the reference's OWN expert selection (oracle/_ref/ifa_ref_moe_rows = HostTensorOpr::BuildRowsForMoE compiled
from /root/reference by `make -C oracle ref_moe_rows`) -> tests/golden/ref_moe_rows.npz: router probability matrices (softmax
outputs as F16, incl. exact ties, probabilities below the 1e-5 cut and fewer experts than top_k) and, per token, the experts and
weights the reference keeps, in its order.  Build container only.

    python tests/golden/gen_moe_rows_fixtures.py
"""
import json, os, struct, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BIN = os.path.join(ROOT, "oracle", "_ref", "ifa_ref_moe_rows")


def run_ref(probs, top_k, norm):
    p = np.ascontiguousarray(probs, np.float16)
    T, E = p.shape
    with tempfile.TemporaryDirectory() as d:
        fi, fo = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fi, "wb") as f:
            f.write(struct.pack("<iiiii", 0x49464d31, T, E, top_k, 1 if norm else 0))
            f.write(p.view(np.uint16).tobytes())
        subprocess.run([BIN, fi, fo], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        raw = np.fromfile(fo, dtype=np.uint8)
    rec = raw.view(np.dtype([("n", "<i4"), ("ew", [("e", "<i4"), ("w", "<f4")], 8)]))
    return rec["n"].copy(), rec["ew"]["e"].copy(), rec["ew"]["w"].copy()


def cases():
    rng = np.random.default_rng(77)
    out = []
    for E, K, norm in [(8, 2, True), (8, 2, False), (4, 2, True), (64, 6, True), (16, 8, False), (3, 4, True), (60, 4, True)]:
        T = 96
        lg = rng.normal(0, 2.5, (T, E)).astype(np.float32)
        p = np.exp(lg - lg.max(axis=1, keepdims=True)); p /= p.sum(axis=1, keepdims=True)
        p = p.astype(np.float16)
        p[0, :] = p[0, 0]                                  # all equal
        p[1, 1] = p[1, 0]; p[2, min(3, E - 1)] = p[2, 0]   # pairs of exact ties
        p[3, :] = np.float16(0.0); p[3, E - 1] = np.float16(1.0)        # one expert only (the rest below the 1e-5 cut)
        p[4, :] = np.float16(5e-6)                         # nothing survives the cut
        out.append(dict(probs=p, top_k=K, norm=norm))
    return out


def main():
    if not os.path.exists(BIN):
        sys.exit("build oracle/_ref/ifa_ref_moe_rows first: make -C oracle ref_moe_rows (needs /root/reference)")
    arrays, meta = {}, []
    for i, c in enumerate(cases()):
        n, e, w = run_ref(c["probs"], c["top_k"], c["norm"])
        arrays["c%d_probs" % i] = c["probs"]; arrays["c%d_n" % i] = n; arrays["c%d_e" % i] = e; arrays["c%d_w" % i] = w
        meta.append(dict(top_k=c["top_k"], norm=bool(c["norm"])))
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_moe_rows.npz"), **arrays)
    print("wrote", len(meta), "cases")


if __name__ == "__main__":
    main()
