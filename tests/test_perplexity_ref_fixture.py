"""The perplexity harness pinned to the REFERENCE's own tool: tests/golden/ref_perplexity.npz holds the PPL, error estimate
and per-window running PPL that src/tools/perplexity.cc (compiled where it lies: `make -C oracle ref_perplexity`) printed
for a tiny llama2.c checkpoint and a 301-token stream (generator: tests/golden/gen_perplexity_fixture.py).
  * CPU: the whole-model oracle's logits through the tool's statistics (windows, float log-softmax, double sums);
  * GPU: host/perplexity.cc through the C ABI (ifa_engine_perplexity) on the same checkpoint."""
import json
import os

import numpy as np
import pytest

from inferflow_amd import dtypes as dt
from tests import engine_fixtures as fx
from tests.model_util import oracle_model_from_host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = np.load(os.path.join(ROOT, "tests", "golden", "ref_perplexity.npz"))
SHAPE = json.loads(str(FX["shape"]))
TOKENS = FX["tokens"].astype(np.int32)
MAXLEN, STRIDE = int(FX["max_length"]), int(FX["stride"])


def _stats(window_logits, windows):
    nll = nll2 = 0.0
    count, running = 0, []
    for lg, win in zip(window_logits, windows):
        lg = lg.astype(np.float32)
        for i in range(len(win) - 1):
            row = lg[i]
            m = row.max()
            v = -(float(row[win[i + 1]] - m) - np.log(float(np.exp(row - m, dtype=np.float32).astype(np.float64).sum())))
            nll += v; nll2 += v * v
        count += len(win) - 1
        running.append(float(np.exp(nll / count)))
    mean = nll / count
    var = nll2 / count - mean * mean
    ppl = float(np.exp(mean))
    return ppl, float(np.sqrt(var / (count - 1))) * ppl, count, running


def test_oracle_logits_reproduce_the_reference_tools_perplexity():
    w = fx.make_weights(SHAPE, int(FX["seed"]), float(FX["std"]), shared_classifier=False)
    om = oracle_model_from_host(fx.host_tensors(w, SHAPE, dt.F16), SHAPE, int(FX["ctx"]), dt.F16, rope_order=1)
    wins = [TOKENS[s:s + MAXLEN] for s in range(0, len(TOKENS), STRIDE)]
    logits = [om.forward(win, 0, nthreads=4)[1] for win in wins]
    ppl, err, count, running = _stats(logits, wins)
    assert count == int(FX["count"])
    # F16 whole-model logits of the restatement vs the reference CPU engine's: the mean nll of 297 tokens to 2e-3 relative
    assert abs(np.log(ppl) - np.log(float(FX["ppl"]))) <= 2e-3 * np.log(float(FX["ppl"])), (ppl, float(FX["ppl"]))
    assert abs(err - float(FX["err"])) <= 0.03 * float(FX["err"])
    for a, b in zip(running, FX["running"]):
        assert abs(np.log(a) - np.log(b)) <= 3e-3 * np.log(b)


@pytest.mark.gpu
def test_engine_perplexity_matches_the_reference_tool(tmp_path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from inferflow_amd.engine import InferenceEngine
    ini, _ = fx.write_model_dir(str(tmp_path), fmt="llama2.c", wd="F16", kvd="F16", ctx=int(FX["ctx"]), s=SHAPE, seed=int(FX["seed"]),
                                std=float(FX["std"]), shared_classifier=False, ret="true")
    eng = InferenceEngine.from_ini(ini)
    ppl, err, count = eng.perplexity(TOKENS, max_length=MAXLEN, stride=STRIDE)
    eng.close()
    assert count == int(FX["count"])
    assert abs(np.log(ppl) - np.log(float(FX["ppl"]))) <= 2e-3 * np.log(float(FX["ppl"])), (ppl, float(FX["ppl"]))
    assert abs(err - float(FX["err"])) <= 0.03 * float(FX["err"]), (err, float(FX["err"]))
