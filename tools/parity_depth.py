#!/usr/bin/env python3
"""How the GPU-vs-oracle logit difference of the T = 1 (int8-activation) decode path grows with depth at Llama-2-7B widths:
the first N layers of the same synthetic model on both sides (GPU: option debug_layers; oracle: an N-layer model with the same
tensors), N = 1, 2, 4, 8, 16, 32.  Prints cosine and max |dlogit| / std of the last prompt row (T = 4 prefill, F16 activations)
and of three teacher-forced decode steps (int8 activations).  A smooth ~sqrt(N) growth is re-quantised rounding noise (the two
sides differ in fp32 summation order); a jump at one depth would be a fault.

    python tools/parity_depth.py [q4|q3h] [f16|q8]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle as o
from inferflow_amd import dtypes as dt, synth
from tests import gpu_util as g

wd = {"q4": dt.Q4_B32T1A, "q3h": dt.Q3H_B64T1}[sys.argv[1] if len(sys.argv) > 1 else "q4"]
kvd = {"f16": dt.F16, "q8": dt.Q8_B32T2}[sys.argv[2] if len(sys.argv) > 2 else "f16"]
max_ctx = 64
wk, host, s = synth.build("llama2_7b", wd, kvd, max_ctx=max_ctx, keep_host=True)
quant = {}
def qt(key):
    if key not in quant:
        target, arr, rows, cols = host[key]
        quant[key] = (target, arr.reshape(rows, cols).view(np.uint16) if target == dt.F16 else o.quantize(target, arr.reshape(rows, cols)), rows, cols)
    return quant[key]

prompt = np.random.default_rng(2024).integers(3, s["vocab"], 4).astype(np.int32)
def cm(a, b):
    a = a.astype(np.float32); b = b.astype(np.float32)
    return float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)), float(np.abs(a - b).max()), float(b.std())

for N in (1, 2, 4, 8, 16, 32):
    om = o.Model(dim=s["dim"], layers=N, heads=s["heads"], kv_heads=s["kv_heads"], head_dim=s["head_dim"], ffn=s["ffn"], vocab=s["vocab"], max_ctx=max_ctx, kv_dtype=kvd)
    for key in host:
        layer, tid = key[0], key[1]
        if layer >= N:
            continue
        target, data, rows, cols = qt(key)
        om.set_tensor(max(layer, 0), tid, target, data, rows, cols)
    wk.set_option("debug_layers", N)
    wk.reset()
    lg = torch.empty((len(prompt), s["vocab"]), dtype=torch.float16, device="cuda")
    # (forward() runs every layer whatever debug_layers says: the prompt goes through the T = 1 path on both sides instead)
    cur = None
    rows = []
    for i, t in enumerate(prompt):
        toks, _ = wk.decode(int(t), i, 1)
        lg_gpu = wk.read_buffer("logits").view(np.float16).copy()
        t_or, l_or = om.forward(np.array([t], np.int32), i)
        rows.append(cm(lg_gpu, l_or[0]))
        cur = int(t_or)
    for i in range(3):
        toks, _ = wk.decode(cur, len(prompt) + i, 1)
        lg_gpu = wk.read_buffer("logits").view(np.float16).copy()
        t_or, l_or = om.forward(np.array([cur], np.int32), len(prompt) + i)
        rows.append(cm(lg_gpu, l_or[0]) + (int(toks[0]) == int(t_or),))
        cur = int(t_or)
    print("layers %2d: " % N + "  ".join("cos %.6f d/std %.4f" % (r[0], r[1] / r[2]) for r in rows), flush=True)
