"""CPU: the oracle's whole-model decoder (oracle/ifa_oracle_model.c, F16 weights, the GPU path's rounding points) against
the fixtures produced by the REFERENCE's own CPU inference path (tests/golden/ref_model_*.npz; generator
tests/golden/gen_model_fixtures.py, recipe oracle/Makefile `ref_engine`).  This is the pin for the model wiring: norm
placement, RoPE pairing of the llama2.c checkpoints (qk_column_order 0), GQA head indexing, residual order, lm_head
(shared and separate classifier), greedy selection incl. the excluded unk id."""
import numpy as np
import pytest

from inferflow_amd import dtypes as dt
from tests import engine_fixtures as fx
from tests import ref_fixtures as rf
from tests.model_util import oracle_model_from_host


@pytest.mark.parametrize("name", rf.names())
def test_oracle_model_matches_reference_cpu_path(name):
    fxt = rf.load(name)
    s = fxt["shape"]
    w = rf.weights(fxt)
    host = fx.host_tensors(w, s, dt.F16)
    om = oracle_model_from_host(host, s, fxt["ctx"], dt.F16, rope_order=rf.rope_order(fxt))
    prompt = fxt["prompt"]
    _, lg = om.forward(prompt, 0, nthreads=4)
    rows = []
    pos = len(prompt)
    n = len(fxt["tokens"])
    for step in range(n - 1):
        _, l1 = om.forward(np.array([fxt["tokens"][step]], np.int32), pos, nthreads=4)   # teacher-forced with the reference's ids
        rows.append(l1[0])
        pos += 1
    st = rf.check_run(fxt, lg, rows, "oracle[%s]" % name)
    assert st["steps"] == n >= 64
    print(name, st)


def test_fixtures_cover_mha_gqa_and_both_classifiers():
    names = rf.names()
    assert len(names) >= 4
    shapes = [rf.load(n) for n in names]
    assert any(f["shape"]["kv_heads"] < f["shape"]["heads"] for f in shapes)
    assert any(f["shape"]["kv_heads"] == f["shape"]["heads"] for f in shapes)
    assert any(f["shared_classifier"] for f in shapes) and any(not f["shared_classifier"] for f in shapes)
    for f in shapes:
        assert len(f["tokens"]) >= 64 and f["prefill_logits"].shape == (len(f["prompt"]), f["shape"]["vocab"])
