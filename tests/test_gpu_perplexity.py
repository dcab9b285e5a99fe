"""-m gpu: the perplexity harness (src/tools/perplexity.cc) through the C ABI and the CLI, against the same
statistics computed from the whole-model oracle's logits."""
import os
import subprocess

import numpy as np
import pytest

from inferflow_amd import build
from inferflow_amd import dtypes as dt
from inferflow_amd.engine import InferenceEngine, EngineError
from tests import engine_fixtures as fx
from tests.model_util import oracle_model_from_host

pytestmark = pytest.mark.gpu


def _oracle_ppl(om, tokens, max_length, stride):
    """perplexity.cc:41-83 (windows), :100-119 (float log-softmax, double sum), :268-276 (estimate)."""
    nll, nll2, count, running = 0.0, 0.0, 0, []
    for start in range(0, len(tokens), stride):
        win = np.asarray(tokens[start:start + max_length], np.int32)
        _, lg = om.forward(win, 0, nthreads=4)
        lg = lg.astype(np.float16).astype(np.float32)
        for i in range(len(win) - 1):
            row = lg[i]
            m = row.max()
            v = -(float(row[win[i + 1]] - m) - float(np.log(np.exp(row - m, dtype=np.float32).astype(np.float64).sum())))
            nll += v; nll2 += v * v
        count += len(win) - 1
        running.append(float(np.exp(nll / count)) if count else 0.0)
    mean = nll / count
    var = nll2 / count - mean * mean
    ppl = float(np.exp(mean))
    return ppl, (float(np.sqrt(var / (count - 1))) * ppl if var > 0 else 0.0), count, running


@pytest.mark.parametrize("wd_name,wd,kv_name,kvd", [("Q4", dt.Q4_B32T1A, "F16", dt.F16), ("Q3H", dt.Q3H_B64T1, "Q8", dt.Q8_B32T2)],
                         ids=["q4_kvf16", "q3h_kvq8"])
def test_perplexity_matches_oracle_statistics(tmp_path, wd_name, wd, kv_name, kvd):
    ini, w = fx.write_model_dir(str(tmp_path), fmt="llama2.c", wd=wd_name, kvd=kv_name, ctx=64)
    eng = InferenceEngine.from_ini(ini)
    s = fx.SHAPE
    om = oracle_model_from_host(fx.host_tensors(w, s, wd), s, 64, kvd, rope_order=1)
    tokens = np.random.default_rng(11).integers(3, s["vocab"], 150).astype(np.int32)
    # overlapping windows (stride < max_length) and a ragged last window, as the reference cuts them
    ppl, err, count = eng.perplexity(tokens, max_length=48, stride=40)
    ppl_o, err_o, count_o, _ = _oracle_ppl(om, tokens, 48, 40)
    assert count == count_o == 47 + 47 + 47 + 29
    # logits agree to the GEMV/attention tolerance (tests/test_gpu_engine.py); the mean nll over 170 tokens to 1e-3 relative
    assert abs(np.log(ppl) - np.log(ppl_o)) <= 1e-3 * np.log(ppl_o), (ppl, ppl_o)
    assert abs(err - err_o) <= 0.02 * err_o + 1e-6, (err, err_o)
    assert eng.query_count() == 0
    # error conventions: a window that does not fit the context, a token out of range
    with pytest.raises(EngineError):
        eng.perplexity(tokens, max_length=64, stride=64)
    with pytest.raises(EngineError):
        eng.perplexity([1, 2, 5000], max_length=16, stride=16)
    eng.close()


def test_perplexity_needs_output_tensors(tmp_path):
    ini, _ = fx.write_model_dir(str(tmp_path), fmt="llama2.c", wd="Q4", kvd="F16", ret="false")
    eng = InferenceEngine.from_ini(ini)
    with pytest.raises(EngineError, match="return_output_tensors"):
        eng.perplexity(list(range(3, 40)), max_length=16, stride=16)
    eng.close()


def test_quantized_formats_rank_by_perplexity_on_the_f16_models_own_text(tmp_path):
    """The tool's purpose: compare formats.  Text = the F16 model's own greedy continuation, so the
    F16 model scores it best and coarser formats score it no better than finer ones (up to noise)."""
    s = fx.SHAPE
    ini16, _ = fx.write_model_dir(str(tmp_path / "f16"), fmt="llama2.c", wd="F16", kvd="F16", ctx=128)
    e16 = InferenceEngine.from_ini(ini16)
    prompt = [1] + list(np.random.default_rng(2).integers(3, s["vocab"], 7))
    qid = e16.add_query(prompt)
    text, _ = e16.generate(qid, 100)
    e16.remove_query(qid)
    tokens = prompt + text
    ppl = {"F16": e16.perplexity(tokens, 108, 108)[0]}
    e16.close()
    for name in ("Q8", "Q4", "Q2"):
        ini, _ = fx.write_model_dir(str(tmp_path / name), fmt="llama2.c", wd=name, kvd="F16", ctx=128)
        e = InferenceEngine.from_ini(ini)
        ppl[name] = e.perplexity(tokens, 108, 108)[0]
        e.close()
    assert ppl["F16"] <= ppl["Q8"] * 1.02 and ppl["Q8"] <= ppl["Q4"] * 1.05 and ppl["Q4"] < ppl["Q2"], ppl
    assert ppl["F16"] < 0.5 * s["vocab"]        # the model predicts its own text far better than chance


def test_perplexity_cli(tmp_path):
    build.build_library()
    cli = os.path.join(os.path.dirname(build.CLI_PATH), "ifa_perplexity")
    ini, _ = fx.write_model_dir(str(tmp_path), fmt="llama2.c", wd="Q4", kvd="F16", ret="false")   # the tool switches it on
    tokens = np.random.default_rng(11).integers(3, fx.SHAPE["vocab"], 100)
    data = tmp_path / "tokens.txt"
    data.write_text(" ".join(str(int(t)) for t in tokens) + "\n")
    ppl_ini = tmp_path / "perplexity.ini"
    ppl_ini.write_text("[main]\ninference_engine_config = ${config_dir}/engine.ini\ntest_data_file = ${config_dir}/tokens.txt\nmax_length = 32\nstride = 32\n")
    out = subprocess.run([cli, str(ppl_ini)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert [l.split("]")[0] for l in lines[:-1]] == ["[0", "[1", "[2", "[3"]
    assert lines[-1].startswith("Final estimate: PPL = ")
    final = float(lines[-1].split("=")[1].split("+/-")[0])
    eng = InferenceEngine.from_ini(fx.write_model_dir(str(tmp_path / "again"), fmt="llama2.c", wd="Q4", kvd="F16")[0])
    assert abs(final - eng.perplexity(tokens, 32, 32)[0]) <= 1e-3 * final
    eng.close()
