#!/usr/bin/env python3
"""MO vs tiled rows GEMM at full-size widths: logits of one batched decode step with rows_mo 1 / 0.
    python tools/debug_rows_mo.py <shape> [queries] [layers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inferflow_amd import dtypes as dt, synth
shape = sys.argv[1] if len(sys.argv) > 1 else "llama2_7b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 2
wk, _, s = synth.build(shape, dt.Q4_B32T1A, dt.F16, max_ctx=64, layers=layers, vocab=8000)
V = s["vocab"]
wk.kv_slots(2 * n)
rng = np.random.default_rng(5)
prompts = [rng.integers(3, V, 3 + i % 5).astype(np.int32) for i in range(n)]
res = {}
for mo in (1, 0):
    wk.set_option("rows_mo", mo)
    cur, pos = [], []
    for i, pr in enumerate(prompts):
        wk.select_kv(mo * n + i)
        cur.append(wk.forward(pr, 0)); pos.append(len(pr))
    lg = torch.empty((n, V), dtype=torch.float16, device="cuda")
    toks = wk.decode_batch(cur, pos, list(range(mo * n, mo * n + n)), lg)
    res[mo] = (lg.cpu().numpy().copy(), [int(t) for t in toks], cur)
a, b = res[1][0].astype(np.float32), res[0][0].astype(np.float32)
print("first tokens equal:", res[1][2] == res[0][2])
print("tokens mo   :", res[1][1]); print("tokens tiled:", res[0][1])
for r in range(n):
    d = np.abs(a[r] - b[r])
    print("row %d: max |d| %.4f  identical %s  nan %d" % (r, d.max(), np.array_equal(res[1][0][r], res[0][0][r]), int(np.isnan(a[r]).sum())))
# batched rows against each query's own single-query step (T = 1 path), and against the op-by-op rows
wk.set_option("rows_mo", 1)
solo, firsts, pos = [], [], []
for i, pr in enumerate(prompts):
    wk.select_kv(i)
    t0 = wk.forward(pr, 0); firsts.append(t0); pos.append(len(pr))
    nxt, _ = wk.decode(t0, len(pr), 1); solo.append(int(nxt[0]))
    wk.forward(pr, 0)
lg1 = torch.empty((n, V), dtype=torch.float16, device="cuda")
got = [int(t) for t in wk.decode_batch(firsts, pos, list(range(n)), lg1)]
wk.set_option("batch_fused", 0)
lg2 = torch.empty((n, V), dtype=torch.float16, device="cuda")
got2 = [int(t) for t in wk.decode_batch(firsts, pos, list(range(n)), lg2)]
print("solo        :", solo); print("batch fused :", got); print("batch op    :", got2)
a, b = lg1.float().cpu().numpy(), lg2.float().cpu().numpy()
for r in range(n):
    print("row %d fused vs op rows: max |d| %.4f cos %.6f" % (r, np.abs(a[r] - b[r]).max(), float((a[r] * b[r]).sum() / (np.linalg.norm(a[r]) * np.linalg.norm(b[r])))))
