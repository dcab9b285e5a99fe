"""Child process of tests/test_gpu_wait_timeout.py (its own process: a timed-out wait switches the waiting launches off for the
whole process).  A split-K prefill product whose first K parts never publish (debug bit 14 of ifa_gemm_big_tiles): the last part's
bounded wait must give up, the synchronising call must FAIL with IFA_ERR_STATE, the waiting launches must be off afterwards and the
same call must then return the unsplit kernel's product; a worker's prompt and steps must work in that state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import inferflow_amd as ia
from inferflow_amd import dtypes as dt, synth
from inferflow_amd._capi import IfaError
from tests import gpu_util as g

L = g.capi()
IFA_ERR_STATE = -4                                       # include/inferflow_amd.h
T, rows, cols = 1024, 4096, 4096
torch.manual_seed(1)
w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
W = g.quantize(dt.Q4_B32T1A, w)
x = (torch.randn(T, cols, device="cuda") * 0.5).half()
assert L.ifa_inlaunch_waits_enabled() == 1
L.ifa_gemm_big_tiles(1 | (1 << 12))                      # split-K off: the product to compare with
y_ref = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x))
ia.check(L.ifa_stream_sync(g.stream()))
L.ifa_gemm_big_tiles(1)
y_split = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x))
ia.check(L.ifa_stream_sync(g.stream()))
assert not np.array_equal(y_split, y_ref)                # (the split-K launch is what runs here: another summation order)
L.ifa_gemm_big_tiles(1 | (1 << 14))                      # the first halves leave without publishing
y = g.gemm(dt.Q4_B32T1A, W, rows, cols, x)
try:
    ia.check(L.ifa_stream_sync(g.stream()))
    print("FAIL: the synchronising call did not report the timed-out wait"); sys.exit(1)
except IfaError as e:
    msg = str(e)
    assert e.code == IFA_ERR_STATE, e.code
    assert "timed out" in msg and "0x81" in msg, msg
assert L.ifa_inlaunch_waits_enabled() == 0               # off for the process
ia.check(L.ifa_stream_sync(g.stream()))                  # the code was consumed: the next synchronisation is clean
y2 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x))      # same call again: no launch waits any more
ia.check(L.ifa_stream_sync(g.stream()))
assert np.array_equal(y2, y_ref)
L.ifa_gemm_big_tiles(1)
y3 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x))
assert np.array_equal(y3, y_ref)
# a worker in that state: prompt (rows GEMM without K parts), captured steps, batched step
wk, _, s = synth.build("test_gqa", dt.Q4_B32T1A, dt.F16, max_ctx=64)
prompt = np.random.default_rng(2).integers(3, s["vocab"], 9).astype(np.int32)
t = wk.forward(prompt, 0)
toks, _ = wk.decode(int(t), len(prompt), 4)
assert len(toks) == 4 and all(0 <= int(v) < s["vocab"] for v in toks)
wk.close()
print("wait timeout ok")
