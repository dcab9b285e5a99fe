#!/usr/bin/env python3
"""One shape of the large-tile prefill GEMM, a few launches (for counter passes): T rows cols [tile-force]."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import inferflow_amd as ia
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
L = ia.lib()
T, rows, cols = (int(v) for v in sys.argv[1:4])
force = int(sys.argv[4]) if len(sys.argv) > 4 else 0
d = dt.Q4_B32T1A
w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
W = g.quantize(d, w); x = (torch.randn(T, cols, device="cuda") * 0.5).half(); st = g.stream(); y = g.empty_f16(T, rows)
L.ifa_gemm_big_tiles(1 | (force << 8))
for _ in range(12):
    ia.check(L.ifa_gemm(d, g.p(W), rows, cols, g.p(x), T, None, g.p(y), st))
torch.cuda.synchronize()
