"""__graft_entry__.smoke(): one small decode of the hot path on cuda:0 (prefill
+ fused graph decode of a tiny Q4 model), checked against the CPU oracle -- the timed path within its tolerance, then the same
steps in the reference kernels' summation order (option exact_order) bit for bit."""
import numpy as np
import torch

import oracle as o
from inferflow_amd import dtypes as dt, synth
from tests import gpu_util as g
from tests.model_util import oracle_model_from_host


def run():
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    wk, host, s = synth.build("test_gqa", dt.Q4_B32T1A, dt.F16, max_ctx=32, quant_threshold=0, std=0.06, keep_host=True)
    ok, why = wk.fused_supported()
    assert ok, why
    om = oracle_model_from_host(host, s, 32, dt.F16)
    prompt = np.array([5, 17, 400, 33, 2], np.int32)
    lg = torch.empty((len(prompt), s["vocab"]), dtype=torch.float16, device="cuda")
    tok = wk.forward(prompt, 0, lg)
    tok_o, lg_o = om.forward(prompt, 0)
    a, b = g.host(lg).astype(np.float32), lg_o.astype(np.float32)
    cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b)))
    assert cos > 0.9995, cos
    toks, ms = wk.decode(tok, len(prompt), 4)
    cur = tok
    for i in range(4):
        t_o, l_o = om.forward(np.array([cur], np.int32), len(prompt) + i)
        top2 = np.sort(l_o[0].astype(np.float32))[-2:]
        assert int(toks[i]) == t_o or top2[1] - top2[0] <= 0.05, "decode step %d mismatch" % i
        cur = int(toks[i])
    # the order-exact step: logits and ids of four single-token steps equal the oracle's, every bit
    wk.set_option("exact_order", 1)
    om.reset()
    cur = int(prompt[0])
    for i in range(4):
        t_gpu, _ = wk.decode(cur, i, 1)
        t_o, l_o = om.forward(np.array([cur], np.int32), i)
        assert np.array_equal(wk.read_buffer("logits").view(np.uint16), l_o[0].view(np.uint16)), "order-exact step %d: logits differ from the oracle's" % i
        assert int(t_gpu[0]) == int(t_o)
        cur = int(t_o)
    wk.close()
