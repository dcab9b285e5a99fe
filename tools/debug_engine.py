import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as o
from inferflow_amd import dtypes as dt, synth
from tests import gpu_util as g
from tests.model_util import oracle_model_from_host

def cos(a, b):
    a = a.astype(np.float32).ravel(); b = b.astype(np.float32).ravel()
    return float((a*b).sum()/(np.linalg.norm(a)*np.linalg.norm(b)+1e-30)), float(np.abs(a-b).max())

wk, host, s = synth.build("test_gqa", dt.Q4_B32T1A, dt.F16, max_ctx=32, quant_threshold=0, std=0.06, keep_host=True, layers=1)
om = oracle_model_from_host(host, s, 32, dt.F16)
prompt = np.array([5], np.int32)
lg = torch.empty((1, s["vocab"]), dtype=torch.float16, device="cuda")
tok = wk.forward(prompt, 0, lg)
tok_o, lg_o = om.forward(prompt, 0)
hid = wk.read_buffer("hidden", 0, nbytes=s["dim"]*2).view(np.float16)
lm = host[(-1, 3)][1].reshape(s["vocab"], s["dim"])
ref = lm.astype(np.float32) @ hid.astype(np.float32)
print("gpu vs ref", cos(g.host(lg)[0], ref), "oracle vs ref", cos(lg_o[0], ref))
y = g.host(g.gemv(dt.F16, g.dev(lm), s["vocab"], s["dim"], g.dev(hid), dt.F16))
print("standalone gemv vs ref", cos(y, ref))
gl = g.host(lg)[0].astype(np.float32)
bad = np.nonzero(np.abs(gl - ref) > 0.05)[0]
print("bad rows", len(bad), bad[:40])
