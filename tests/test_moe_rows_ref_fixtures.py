"""The per-row expert selection of a mixture-of-experts layer against rows of the REFERENCE itself
(tests/golden/ref_moe_rows.npz, produced by tests/golden/gen_moe_rows_fixtures.py from HostTensorOpr::BuildRowsForMoE,
src/tensor/host_tensor_opr.cc:190-244, compiled where it lies): the oracle's restatement on the CPU, the device routing
kernel (ifa_moe_route_topk, through the C ABI) on the GPU."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as o

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_moe_rows.npz")
Z = np.load(FIX)
META = json.loads(bytes(Z["meta"]).decode())


@pytest.mark.parametrize("i", range(len(META)))
def test_oracle_selection_is_the_reference_selection(i):
    m, probs, n_ref, e_ref, w_ref = META[i], Z["c%d_probs" % i], Z["c%d_n" % i], Z["c%d_e" % i], Z["c%d_w" % i]
    for t in range(probs.shape[0]):
        idx, w = o.moe_topk(probs[t].astype(np.float32), m["top_k"], m["norm"])
        n = int(n_ref[t])
        assert len(idx) == n, (t, idx, n)
        assert [int(v) for v in idx] == [int(v) for v in e_ref[t][:n]], t          # same experts in the same (descending) order
        assert np.array_equal(np.asarray(w, np.float32).view(np.uint32), w_ref[t][:n].view(np.uint32)), t       # bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(META)))
def test_device_routing_is_the_reference_selection(i):
    import torch
    import inferflow_amd as ia
    from tests import gpu_util as g
    m, probs, n_ref, e_ref, w_ref = META[i], Z["c%d_probs" % i], Z["c%d_n" % i], Z["c%d_e" % i], Z["c%d_w" % i]
    T, E = probs.shape
    k = min(m["top_k"], 8)
    sel = torch.zeros((T, k), dtype=torch.int32, device="cuda")
    wts = torch.zeros((T, k), dtype=torch.float16, device="cuda")
    ia.check(g.capi().ifa_moe_route_topk(g.p(g.dev(probs)), T, E, k, 1 if m["norm"] else 0, g.p(sel), g.p(wts), g.stream()))
    sel_h, w_h = sel.cpu().numpy(), g.host(wts)
    for t in range(T):
        n = int(n_ref[t])
        order = np.argsort(e_ref[t][:n], kind="stable")           # the C ABI lists a row's experts in ascending id order
        exp_sel = [int(e_ref[t][j]) for j in order] + [-1] * (k - n)
        exp_w = [np.float16(w_ref[t][j]) for j in order] + [np.float16(0)] * (k - n)
        assert sel_h[t].tolist() == exp_sel, t
        assert np.array_equal(w_h[t].view(np.uint16), np.array(exp_w, np.float16).view(np.uint16)), t
