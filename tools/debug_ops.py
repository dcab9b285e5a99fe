import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as o
import inferflow_amd as ia
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
rng = np.random.default_rng(0)
cols = 4096
x = rng.normal(0, 1.0, (3, cols)).astype(np.float16)
w = rng.normal(1, 0.1, cols).astype(np.float16)
y = g.empty_f16(3, cols)
ia.check(g.capi().ifa_layernorm(0, g.p(g.dev(x)), 3, cols, g.p(g.dev(w)), None, 0.0, 1e-5, g.p(y), g.stream()))
g.sync()
exp = o.rmsnorm(x, w); got = g.host(y)
d = g.half_ulp_diff(got, exp)
print("rms: nmismatch", (d != 0).sum(), "max ulp", d.max(), "first", got.ravel()[:4], exp.ravel()[:4])
a = rng.normal(0, 1.0, (4, 512)).astype(np.float16); b = rng.normal(0, 1.0, (4, 512)).astype(np.float16)
c = g.empty_f16(4, 512)
ad, bd = g.dev(a), g.dev(b)
for name, fn, e in [("add", lambda: g.capi().ifa_add(g.p(ad), g.p(bd), a.size, 0, g.p(c), g.stream()), o.add(a, b)),
                    ("mul", lambda: g.capi().ifa_mul(g.p(ad), g.p(bd), a.size, g.p(c), g.stream()), o.mul(a, b)),
                    ("scale", lambda: g.capi().ifa_scale(g.p(ad), 0.37, a.size, g.p(c), g.stream()), o.scale(a, 0.37))]:
    ia.check(fn()); g.sync()
    d = g.half_ulp_diff(g.host(c), e)
    print(name, "nmismatch", (d != 0).sum(), "max ulp", d.max())
# attention
heads, kv_heads, hd, n_ctx, qt = 8, 8, 64, 37, 1
q = rng.normal(0, 1.0, (qt, heads, hd)).astype(np.float16)
k = rng.normal(0, 1.0, (n_ctx, kv_heads * hd)).astype(np.float16)
v = rng.normal(0, 1.0, (n_ctx, kv_heads * hd)).astype(np.float16)
exp = o.attention(q, k, v, dt.F16, n_ctx, n_ctx - qt, heads, kv_heads, hd, 2.0)
out = g.empty_f16(qt, heads * hd)
qd, kd, vd = g.dev(q), g.dev(k), g.dev(v)
ia.check(g.capi().ifa_attention(g.p(qd), g.p(kd), g.p(vd), dt.F16, n_ctx, qt, n_ctx - qt, heads, kv_heads, hd, 2.0, 0, 0, heads, g.p(out), g.stream()))
g.sync()
got = g.host(out).astype(np.float32)
print("attn nan count", np.isnan(got).sum(), "maxdiff", np.nanmax(np.abs(got - exp.astype(np.float32))))
