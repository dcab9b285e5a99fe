"""-m gpu: the per-phase key space of the reference's InferencePerfStat (GpuInferenceWorker::UpdatePerfStat,
src/transformer/inference_worker.cc:2670-2697; keys (layer + 1) * 10000 + phase, :296-322, 806-950, 1030-1400, 1750-1880) behind
worker option perf_stat / ifa_model_perf_stat and, in study mode, behind InferenceEngine::Infer's InferenceResult::perf_stat."""
import numpy as np
import pytest

from inferflow_amd import dtypes as dt, synth
from tests import engine_fixtures as fx

pytestmark = pytest.mark.gpu

LAYER0 = [10000, 10010, 10030, 10050, 10060, 10090, 10300, 10700, 10710, 10730, 10750, 10760, 10780, 10800]


def test_worker_perf_stat_fills_the_reference_key_space_and_changes_no_token():
    wk, _, s = synth.build("test_gqa", dt.Q4_B32T1A, dt.F16, max_ctx=64)
    ref, _, _ = synth.build("test_gqa", dt.Q4_B32T1A, dt.F16, max_ctx=64)
    ref.set_option("fused", 0); ref.set_option("batch_fused", 0)      # the op-by-op step and prompt: what perf_stat times
    L = s["layers"]
    prompt = np.random.default_rng(2).integers(3, s["vocab"], 7).astype(np.int32)
    wk.set_option("perf_stat", 1)
    t = wk.forward(prompt, 0); tr = ref.forward(prompt, 0)
    assert int(t) == int(tr)
    st = wk.perf_stat(clear=False)
    want = set(LAYER0) | {(l + 1) * 10000 for l in range(min(L, 6))} | {1, 1000009}
    assert set(st) == want, sorted(set(st) ^ want)
    assert all(v > 0.0 for v in st.values()), st
    # nesting: a whole contains its parts (device times of spans on one stream)
    attn_parts = sum(st[k] for k in (10010, 10030, 10050, 10060, 10090))
    ffn_parts = sum(st[k] for k in (10710, 10730, 10750, 10760, 10780))
    assert st[10300] >= 0.9 * attn_parts and st[10700] >= 0.9 * ffn_parts
    assert st[10000] >= 0.9 * (st[10300] + st[10700] + st[10800])
    # steps ADD to the keys (UpdatePerfStat: iter->second += value); tokens equal the op-by-op step's
    toks, _ = wk.decode(int(t), len(prompt), 3)
    toks_r, _ = ref.decode(int(tr), len(prompt), 3)
    assert list(toks) == list(toks_r)
    st2 = wk.perf_stat(clear=True)
    assert set(st2) == want and all(st2[k] > st[k] for k in want)
    assert wk.perf_stat() == {}                      # cleared
    # off again: the fused step, nothing recorded
    wk.set_option("perf_stat", 0)
    wk.decode(int(toks[-1]), len(prompt) + 3, 2)
    assert wk.perf_stat() == {}
    wk.close(); ref.close()


def test_engine_study_mode_returns_the_phase_keys(tmp_path):
    from inferflow_amd.engine import InferenceEngine
    ini, _ = fx.write_model_dir(str(tmp_path), fmt="llama2.c", wd="Q4", kvd="F16")
    prompt = np.random.default_rng(4).integers(3, 1000, 6).astype(np.int32)
    plain = InferenceEngine.from_ini(ini)
    q = plain.add_query(prompt)
    (_, tok0), = plain.infer()
    assert set(plain.perf_stat()) == {0}             # key 0 (E2E) only, inference_engine.cc:986-988
    plain.close()
    txt = open(ini).read()
    assert "is_study_mode" not in txt
    study_ini = str(tmp_path / "study.ini")
    open(study_ini, "w").write(txt.replace("[transformer_engine]", "[transformer_engine]\nis_study_mode = true", 1))
    eng = InferenceEngine.from_ini(study_ini)
    q = eng.add_query(prompt)
    (_, t0), = eng.infer()
    st = eng.perf_stat()
    assert set(LAYER0) <= set(st) and {0, 1, 1000009} <= set(st), sorted(st)
    assert st[0] >= st[10000] > 0.0
    eng.commit({q: t0}); (_, t1), = eng.infer()
    st1 = eng.perf_stat()
    assert set(st1) == set(st)                       # per Infer() call: the map of THAT step
    assert 0 <= t0 < 1000 and 0 <= t1 < 1000 and 0 <= tok0 < 1000
    eng.close()
