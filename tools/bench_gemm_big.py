#!/usr/bin/env python3
"""Large-T GEMM (GPU box): the in-tree large-tile MFMA kernel vs torch.matmul (hipBLASLt) on the dequantised operand -- a
measurement aid only, the library itself neither links nor loads a vendor GEMM -- with a correctness check of one against the
other.  One JSON line per shape."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import inferflow_amd as ia
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
L = ia.lib()
shapes = [(1024, 4096, 4096), (1024, 11008, 4096), (1024, 4096, 11008), (256, 4096, 4096), (512, 11008, 4096), (4096, 4096, 4096), (1000, 4096, 4096)]
if os.environ.get("IFA_GEMM_SHAPES"):      # "T,rows,cols;T,rows,cols"
    shapes = [tuple(int(v) for v in sh.split(",")) for sh in os.environ["IFA_GEMM_SHAPES"].split(";")]
dts = [dt.Q4_B32T1A, dt.Q3H_B64T1, dt.F16] if "--all" in sys.argv else [dt.Q4_B32T1A]
for d in dts:
    for T, rows, cols in shapes:
        w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
        W = w if d == dt.F16 else g.quantize(d, w)
        x = (torch.randn(T, cols, device="cuda") * 0.5).half()
        st = g.stream()
        res = {"dtype": dt.NAMES[d], "T": T, "rows": rows, "cols": cols}
        outs = {}
        variants = [("own_big", 1, 0), ("own_small", 0, 0), ("library", 0, 2)]
        wdq = w if d == dt.F16 else g.dequantize(d, W, cols)          # F16 operand of the library product
        if "--tiles" in sys.argv:      # force a tile shape: 256 x 256, 128 x 256, 128 x 128
            variants += [("big_256x256", 1 | (1 << 8), 0), ("big_128x256", 1 | (2 << 8), 0), ("big_128x128", 1 | (3 << 8), 0)]
        for name, big, lib in variants:
            L.ifa_gemm_big_tiles(big)
            y = g.empty_f16(T, rows)
            if lib:
                def fn():
                    torch.matmul(x, wdq.t(), out=y)
            else:
                fn = lambda: ia.check(L.ifa_gemm(d, g.p(W), rows, cols, g.p(x), T, None, g.p(y), st))
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / n
            res[name + "_us"] = round(t * 1e6, 1); res[name + "_TFLOPs"] = round(2.0 * T * rows * cols / t / 1e12, 1)
            outs[name] = g.host(y).astype(np.float32)
        L.ifa_gemm_big_tiles(1)
        ref = outs.get("library", outs["own_small"])
        res["max_abs_diff_vs_ref"] = float(np.abs(outs["own_big"] - ref).max()); res["ref_mean_abs"] = float(np.abs(ref).mean())
        print(json.dumps(res), flush=True)
