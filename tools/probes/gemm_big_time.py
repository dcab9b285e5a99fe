#!/usr/bin/env python3
"""Time one shape of the large-tile prefill GEMM (ifa_gemm, Q4_B32T1A): T rows cols [T rows cols ...] -> us per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import inferflow_amd as ia
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
L = ia.lib()
a = [int(v) for v in sys.argv[1:]]
d = dt.Q4_B32T1A
out = []
for i in range(0, len(a), 3):
    T, rows, cols = a[i:i + 3]
    w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
    W = g.quantize(d, w); x = (torch.randn(T, cols, device="cuda") * 0.5).half(); st = g.stream(); y = g.empty_f16(T, rows)
    for _ in range(5): ia.check(L.ifa_gemm(d, g.p(W), rows, cols, g.p(x), T, None, g.p(y), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): ia.check(L.ifa_gemm(d, g.p(W), rows, cols, g.p(x), T, None, g.p(y), st))
    e1.record(); torch.cuda.synchronize()
    out.append("%dx%dx%d %.1f us" % (T, rows, cols, e0.elapsed_time(e1) * 1e3 / n))
print(os.environ.get("IFA_LIB", "default").split("/")[-2] if os.environ.get("IFA_LIB") else "default", " | ".join(out), flush=True)
