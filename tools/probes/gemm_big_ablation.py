import json, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import inferflow_amd as ia
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
L = ia.lib()
L.ifa_gemm_library_min_tokens(0)
d = dt.Q4_B32T1A
for T, rows, cols in [(4096, 4096, 4096), (1024, 4096, 4096)]:
    w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
    W = g.quantize(d, w); x = (torch.randn(T, cols, device="cuda") * 0.5).half(); st = g.stream(); y = g.empty_f16(T, rows)
    for name, mode in (("full", 1), ("no_dequant", 1 | (1 << 4)), ("no_mfma", 1 | (2 << 4)), ("neither", 1 | (3 << 4))):
        L.ifa_gemm_big_tiles(mode)
        fn = lambda: ia.check(L.ifa_gemm(d, g.p(W), rows, cols, g.p(x), T, None, g.p(y), st))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(T, rows, cols, name, round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us", flush=True)
L.ifa_gemm_big_tiles(0)
