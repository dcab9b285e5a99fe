"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0,
checked against the CPU oracle."""
import numpy as np
import torch

import oracle as o
from inferflow_amd import dtypes as dt
from tests import gpu_util as g


def run():
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    rng = np.random.default_rng(0)
    rows, cols = 256, 4096
    w = rng.normal(0, 0.02, (rows, cols)).astype(np.float16)
    x = rng.normal(0, 1.0, (1, cols)).astype(np.float16)
    Wq_gpu = g.quantize(dt.Q4_B32T1A, g.dev(w))
    assert np.array_equal(g.host(Wq_gpu), o.quantize(dt.Q4_B32T1A, w)), "weight quantizer mismatch"
    xq_gpu = g.quantize_act(g.dev(x))
    assert np.array_equal(g.host(xq_gpu), o.quantize_act_q8(x)), "activation quantizer mismatch"
    y = g.host(g.gemv(dt.Q4_B32T1A, Wq_gpu, rows, cols, xq_gpu, dt.Q8_B32T2))
    y_orc = o.gemv_ax8(dt.Q4_B32T1A, g.host(Wq_gpu), rows, cols, g.host(xq_gpu))
    assert g.half_ulp_diff(y, y_orc).max() <= 1, "gemv mismatch"
