#!/usr/bin/env python3
"""Prefill GEMM micro-benchmark (GPU box): TFLOP/s of ifa_gemm at Llama-2-7B shapes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import inferflow_amd as ia
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
L = ia.lib()
import ctypes
for d in (dt.Q4_B32T1A, dt.Q3H_B64T1, dt.F16):
    for T, rows, cols in [(16, 4096, 4096), (128, 4096, 4096), (1024, 4096, 4096), (1024, 11008, 4096), (1024, 4096, 11008), (256, 4096, 4096), (256, 11008, 4096), (512, 4096, 4096), (512, 11008, 4096), (4096, 4096, 4096)]:
        w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
        W = w if d == dt.F16 else g.quantize(d, w)
        x = torch.randn(T, cols, device="cuda").half()
        y = g.empty_f16(T, rows)
        st = g.stream()
        fn = lambda: ia.check(L.ifa_gemm(d, g.p(W), rows, cols, g.p(x), T, None, g.p(y), st))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / n
        print(json.dumps({"dtype": dt.NAMES[d], "T": T, "rows": rows, "cols": cols, "us": t * 1e6,
                          "TFLOPs": 2.0 * T * rows * cols / t / 1e12, "tok_per_s_this_layer": T / t}), flush=True)
