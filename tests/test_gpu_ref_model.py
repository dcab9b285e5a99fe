"""-m gpu: the HIP engine (C++ InferenceEngine facade -> C ABI -> fused decode kernels / prefill kernels) on the SAME
llama2.c checkpoints the REFERENCE's CPU inference path was run on, against the reference's own outputs
(tests/golden/ref_model_*.npz, generator tests/golden/gen_model_fixtures.py).  F16 weights and F16 KV cache: the unquantised
model, so every difference is rounding (tests/ref_fixtures.py states the tolerance).  The st_* fixtures are SAFETENSORS
directories the reference engine itself loaded (HF names, config.json, qk_column_order 2 and 0): they pin
host/model_loader.cc's name map and the RoPE pairing it selects to the reference, not to the oracle."""
import numpy as np
import pytest

from inferflow_amd.engine import InferenceEngine
from tests import engine_fixtures as fx
from tests import ref_fixtures as rf

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", rf.names())
def test_hip_engine_matches_reference_cpu_path(tmp_path, name):
    fxt = rf.load(name)
    s = fxt["shape"]
    ini, _ = rf.write_model_dir(str(tmp_path), fxt, wd="F16", kvd="F16")     # llama2.c or safetensors, as the reference read it
    eng = InferenceEngine.from_ini(ini)
    prompt = fxt["prompt"]
    qid = eng.add_query(prompt)
    assert qid > 0
    (q, tok), = eng.infer()                      # step 0: prefill of the whole prompt, full logits tensor
    lg = eng.last_logits(qid)
    assert lg.shape == (len(prompt), s["vocab"])
    # the engine's own greedy id is the argmax of its logits over the allowed ids (unk excluded like GetSortedTopK)
    assert tok == rf.masked_argmax(lg[-1], fxt["excluded_ids"])
    rows = []
    n = len(fxt["tokens"])
    for step in range(n - 1):
        assert eng.commit({qid: int(fxt["tokens"][step])})       # teacher-forced with the reference's ids
        (q, tok), = eng.infer()                                   # fused decode step (one hipGraph replay)
        row = eng.last_logits(qid)
        assert row.shape == (1, s["vocab"])
        assert tok == rf.masked_argmax(row[0], fxt["excluded_ids"])
        rows.append(row[0].copy())
    st = rf.check_run(fxt, lg, rows, "hip[%s]" % name)
    assert st["steps"] == n >= 64
    eng.close()


@pytest.mark.parametrize("name", ["gqa", "gqa_deep", "st_gqa_hf"])
def test_hip_engine_free_running_greedy_follows_reference(tmp_path, name):
    """Generate() (tokens fed back on the device, no host in the loop) reproduces the reference's greedy ids until the
    first step whose top-2 gap is a tie at this precision."""
    fxt = rf.load(name)
    ini, _ = rf.write_model_dir(str(tmp_path), fxt, wd="F16", kvd="F16", ret="false")
    eng = InferenceEngine.from_ini(ini)
    qid = eng.add_query(fxt["prompt"])
    (q, tok), = eng.infer()
    ref = fxt["tokens"]
    first_tie = int(np.argmax(fxt["top2_gap"] <= rf.LOGIT_TOL)) if (fxt["top2_gap"] <= rf.LOGIT_TOL).any() else len(ref)
    assert first_tie >= 8, "fixture has an early near-tie; pick another seed"
    assert tok == int(ref[0])
    eng.commit({qid: tok})
    gen, _ = eng.generate(qid, first_tie - 1)
    assert list(gen) == [int(t) for t in ref[1:first_tie]]
    eng.close()
