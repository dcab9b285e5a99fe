import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as o
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
def cos(a, b):
    a = a.astype(np.float32).ravel(); b = b.astype(np.float32).ravel()
    return float((a*b).sum()/(np.linalg.norm(a)*np.linalg.norm(b)+1e-30)), float(np.abs(a-b).max())
rng = np.random.default_rng(0)
for rows, cols in [(19,256),(1000,256),(64,256),(65,256),(256,256),(257,256), (1000, 512), (1000,4096)]:
    w = rng.normal(0, 0.06, (rows, cols)).astype(np.float16)
    x = rng.normal(0, 1.0, cols).astype(np.float16)
    ref = w.astype(np.float32) @ x.astype(np.float32)
    y = g.host(g.gemv(dt.F16, g.dev(w), rows, cols, g.dev(x), dt.F16))
    b = np.zeros(rows, np.float16)
    yb = g.host(g.gemv(dt.F16, g.dev(w), rows, cols, g.dev(x), dt.F16, g.dev(b)))
    bad = np.nonzero(np.abs(y.astype(np.float32) - ref) > 0.05)[0]
    print(rows, cols, "nobias", cos(y, ref), "bias0", cos(yb, ref), "nbad", len(bad), bad[:10], flush=True)
