"""Phase times of k_attention_mfma (longest query tile of head 0) + whole-kernel time at a 1024-token prefill."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import inferflow_amd as ia
from inferflow_amd import dtypes as dt
from tests import gpu_util as g
L = ia.lib()
L.ifa_debug_attn_trace.argtypes = [C.c_void_p]
T, heads, hd = int(os.environ.get("T", "1024")), 32, 128
if "KEYS" in os.environ: L.ifa_attention_two_pass_min_keys(int(os.environ["KEYS"]))
if "QMIN" in os.environ: L.ifa_attention_two_pass_min(int(os.environ["QMIN"]))
q = (torch.randn(T, heads * hd, device="cuda") * 0.5).half()
kc = (torch.randn(T, heads * hd, device="cuda") * 0.5).half()
vc = (torch.randn(T, heads * hd, device="cuda") * 0.5).half()
out = g.empty_f16(T, heads * hd)
fn = lambda: ia.check(L.ifa_attention(g.p(q), g.p(kc), g.p(vc), dt.F16, T, T, 0, heads, heads, hd, 1.0, 0, 0, heads, g.p(out), g.stream()))
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
tr = (C.c_ulonglong * 8)()
L.ifa_debug_attn_trace(tr)
t = [tr[i] for i in range(4)]
print("kernel %.1f us; longest tile: QK %.2f us, softmax %.2f us, PV %.2f us (100 MHz ticks)" % (
    e0.elapsed_time(e1) * 100, (t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0, (t[3] - t[2]) / 100.0))
