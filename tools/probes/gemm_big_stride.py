#!/usr/bin/env python3
"""Is the activation-tile fetch of the prefill GEMM sensitive to the row stride (L2 channel camping)?  Same tile, K varied."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import inferflow_amd as ia
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
L = ia.lib()
d = dt.Q4_B32T1A
for T, rows, cols in [(4096, 4096, 4096), (4096, 4096, 4160), (4096, 4096, 4224), (4096, 4096, 4352), (1024, 4096, 4096), (1024, 4096, 4224)]:
    w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
    W = g.quantize(d, w); x = (torch.randn(T, cols, device="cuda") * 0.5).half(); st = g.stream(); y = g.empty_f16(T, rows)
    out = []
    for name, mode in (("auto", 1), ("256x256", 1 | (1 << 8)), ("128x128", 1 | (3 << 8)), ("small", 0)):
        L.ifa_gemm_big_tiles(mode)
        fn = lambda: ia.check(L.ifa_gemm(d, g.p(W), rows, cols, g.p(x), T, None, g.p(y), st))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        out.append("%s %.1f us %.0f TF" % (name, t * 1e6, 2.0 * T * rows * cols / t / 1e12))
    print(T, rows, cols, " | ".join(out), flush=True)
L.ifa_gemm_big_tiles(0)
