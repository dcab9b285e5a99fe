#!/usr/bin/env python3
"""us per launch of ifa_gemm_rows_q4 on Llama-2-7B matrix shapes for 2..16 rows; IFA_ROWS_KERNEL=fdot selects the fdot2
kernel (2..8 rows) instead of the matrix-core one."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import inferflow_amd as ia
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
L = ia.lib()
L.ifa_gemm_rows_q4.restype = C.c_int
L.ifa_gemm_rows_q4.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
print("kernel:", os.environ.get("IFA_ROWS_KERNEL", "mfma"))
for rows, cols in [(4096, 4096), (12288, 4096), (11008, 4096), (4096, 11008), (32000, 4096)]:
    nW = 12                                   # rotate over distinct weights: no cache help
    Ws = []
    for i in range(nW):
        w16 = (torch.randn(rows, cols, device="cuda") * 0.05).half()
        Ws.append(g.repack(dt.Q4_B32T1A, g.quantize(dt.Q4_B32T1A, w16), rows, cols))
    line = "%6d x %6d (%5.1f MB):" % (rows, cols, rows * cols * 0.625 / 1e6)
    for T in (2, 4, 8, 16):
        x = (torch.randn(T, cols, device="cuda")).half()
        y = g.empty_f16(T, rows)
        rc = L.ifa_gemm_rows_q4(g.p(Ws[0]), rows, cols, g.p(x), T, None, g.p(y), g.stream())
        if rc != 0:
            line += "  T=%d n/a" % T
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 60
        e0.record()
        for i in range(n):
            L.ifa_gemm_rows_q4(g.p(Ws[i % len(Ws)]), rows, cols, g.p(x), T, None, g.p(y), g.stream())
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        line += "  T=%d %.1f us (%.2f TB/s)" % (T, us, rows * cols * 0.625 / us / 1e6)
    print(line, flush=True)
