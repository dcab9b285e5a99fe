"""CPU: the decoding strategies against draws of the REFERENCE itself (tests/golden/ref_sampling.npz, produced by
tests/golden/gen_sampling_fixtures.py from oracle/_ref/ifa_ref_sampling = the reference's sampling_strategy.cc compiled
where it lies): both the Python restatement (oracle/sampling.py) and the product's host sampler
(inferflow_amd/host/sampling_strategy.cc through the C ABI) must select the same token, draw after draw."""
import json
import os

import numpy as np
import pytest

from inferflow_amd import engine as E
from oracle import sampling as S

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sampling.npz")
Z = np.load(FIX)
META = json.loads(bytes(Z["meta"]).decode())
STD_FAMILY = (S.STD, S.GREEDY, S.TOP_K, S.TOP_P)


def _case(i):
    m = META[i]
    return m, Z["c%d_logits" % i], [int(t) for t in Z["c%d_text" % i]], Z["c%d_ids" % i], Z["c%d_w" % i], Z["c%d_pool_ids" % i], Z["c%d_pool_w" % i]


def _ids(i):
    m = META[i]
    return "%d-s%d-T%g-v%d" % (i, m["strategy"], m["temperature"], Z["c%d_logits" % i].size)


@pytest.mark.parametrize("i", range(len(META)), ids=[_ids(i) for i in range(len(META))])
def test_restatement_reproduces_the_reference_draws(i):
    m, lg, text, ids, w, pool_ids, pool_w = _case(i)
    cfg = m["config"]
    r = S.JavaRandom(m["seed"])
    st, mu = S.FsdState(), None
    for d in range(m["n_draws"]):
        if m["strategy"] in STD_FAMILY:
            (tok, p), cut = S.choose_tokens(lg, m["strategy"], r, max_k=cfg.get("max_k", 8), top_p=cfg.get("top_p", 0.9),
                                            pool_size=cfg.get("pool_size", 50), temperature=m["temperature"])
        elif m["strategy"] in (S.FSD, S.RANDOM_FSD):
            (tok, p), cut = S.choose_tokens_fsd(lg, m["strategy"], r, st, text, temperature=m["temperature"])
        else:
            (tok, p), cut, mu = S.choose_tokens_ex(lg, m["strategy"], r, temperature=m["temperature"], mu=mu)
        assert tok == int(ids[d]), (d, tok, int(ids[d]))
        assert abs(float(p) - float(w[d])) <= 3e-6 * max(1.0, abs(float(w[d]))), d
    assert [c for c, _ in cut] == [int(t) for t in pool_ids]
    assert np.allclose([float(x) for _, x in cut], pool_w, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("i", range(len(META)), ids=[_ids(i) for i in range(len(META))])
def test_host_sampler_reproduces_the_reference_draws(i):
    m, lg, text, ids, w, pool_ids, pool_w = _case(i)
    cfg = m["config"]
    n = m["n_draws"]
    if m["strategy"] in STD_FAMILY:
        got, probs, pid, ppr = E.sampling_choose(lg, m["strategy"], max_k=cfg.get("max_k", 8), top_p=cfg.get("top_p", 0.9),
                                                 pool_size=cfg.get("pool_size", 50), temperature=m["temperature"], seed=m["seed"], n_draws=n)
    elif m["strategy"] in (S.FSD, S.RANDOM_FSD):
        got, probs, pid, ppr, _ = E.sampling_choose_ex(lg, m["strategy"], temperature=m["temperature"], seed=m["seed"], n_draws=n, text=text, top_p=0.93)
    else:
        got, probs, pid, ppr, _ = E.sampling_choose_ex(lg, m["strategy"], temperature=m["temperature"], seed=m["seed"], n_draws=n)
    assert got == [int(t) for t in ids]
    assert np.allclose(probs, w, rtol=3e-6, atol=3e-7)
    if m["strategy"] != S.MIROSTAT:                  # (the C entry point returns Mirostat's pool before the last mu update)
        assert pid == [int(t) for t in pool_ids]
        assert np.allclose(ppr, pool_w, rtol=2e-5, atol=1e-7)
