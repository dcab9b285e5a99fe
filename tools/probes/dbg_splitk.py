import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
L = g.capi()
T, rows, cols = 1024, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 11008
torch.manual_seed(T)
w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
W = g.quantize(dt.Q4_B32T1A, w)
x = (torch.randn(T, cols, device="cuda") * 0.5).half()
L.ifa_gemm_big_tiles(1)
ya = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x)).astype(np.float32)
ya2 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x)).astype(np.float32)
L.ifa_gemm_big_tiles(1 | (1 << 12))
yb = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x)).astype(np.float32)
L.ifa_gemm_big_tiles(1)
d = np.abs(ya - yb)
print("split vs no split: max diff %.4f, mean |y| %.4f, frac > 0.01: %.4f; split twice identical: %s" % (d.max(), np.abs(yb).mean(), (d > 0.01).mean(), np.array_equal(ya, ya2)))
bad = d > 0.01
print("bad by token row (first 40):", bad.mean(axis=1)[:40].round(2))
print("bad by column block of 128 (first 32):", bad.reshape(T, rows // 128, 128).mean(axis=(0, 2))[:32].round(2))
# is the split result = only one half?
wdq = g.dequantize(dt.Q4_B32T1A, W, cols).float()
h = cols // 2
y1 = g.host((x[:, :h].float() @ wdq[:, :h].t()).contiguous()); y2 = g.host((x[:, h:].float() @ wdq[:, h:].t()).contiguous())
for name, ref in (("first half", y1), ("second half", y2), ("both", y1 + y2)):
    print(name, "max diff to split result %.4f" % np.abs(ya - ref).max(), " rows 0..3:", np.abs(ya - ref).max(axis=1)[:4].round(3))
