"""CPU: the host-side decoding strategies (inferflow_amd/host/sampling_strategy.cc through the C ABI, no GPU involved)
against the Python restatement in oracle/sampling.py and the published java.util.Random known answers."""
import numpy as np
import pytest

from inferflow_amd import engine as E
from oracle import sampling as S


def test_generator_is_the_java_util_random_lcg():
    r = S.JavaRandom(42)
    assert [r.next(32), r.next(32)] == [-1170105035, 234785527]          # new java.util.Random(42).nextInt() x2
    got = E.random_doubles(42, 5)
    r = S.JavaRandom(42)
    assert got == [r.next_double() for _ in range(5)]
    assert got[0] == 0.7275636800328681                                   # new java.util.Random(42).nextDouble()
    r = S.JavaRandom(-7 & ((1 << 64) - 1))                                # negative seeds: (uint64_t)(int64_t)seed
    assert E.random_doubles(-7, 3) == [r.next_double() for _ in range(3)]


@pytest.mark.parametrize("strategy", [S.STD, S.GREEDY, S.TOP_K, S.TOP_P])
@pytest.mark.parametrize("temperature", [1.0, 0.7, 0.0005])
def test_choose_tokens_matches_the_restated_reference(strategy, temperature):
    rng = np.random.default_rng(100 * strategy + int(temperature * 10))
    for case in range(6):
        vocab = [1000, 37, 5, 32000, 64, 1][case]
        logits = (rng.normal(0, 2.0, vocab)).astype(np.float16)
        if case == 2:
            logits[:] = logits[0]                                       # all equal: ties go to the lower id
        if case == 4:
            logits[10:20] = np.float16(3.5)                             # a plateau of equal best values
        seed = 1234 + case
        ids, probs, pool_ids, pool_probs = E.sampling_choose(logits, strategy, max_k=8, top_p=0.9, pool_size=50,
                                                             temperature=temperature, seed=seed, n_draws=20)
        r = S.JavaRandom(seed)
        for d in range(20):
            (tok, p), cut = S.choose_tokens(logits, strategy, r, max_k=8, top_p=0.9, pool_size=50, temperature=temperature)
            assert ids[d] == tok, (case, d)
            assert abs(probs[d] - float(p)) <= 2e-6 * max(1.0, float(p))
        assert pool_ids == [i for i, _ in cut]
        assert np.allclose(pool_probs, [float(w) for _, w in cut], rtol=1e-5, atol=1e-7)
        if strategy == S.GREEDY:
            assert pool_ids == [int(np.flatnonzero(logits == logits.max())[0])] and len(set(ids)) == 1


def test_pool_cuts_follow_top_p_and_max_k():
    logits = np.array([5.0, 4.0, 3.0, 2.0, 1.0, 0.0, -1.0, -2.0, -3.0, -4.0], np.float16)
    _, _, pool, probs = E.sampling_choose(logits, S.TOP_K, max_k=3, top_p=0.5, pool_size=50)
    assert pool == [0, 1, 2]                                            # top_k ignores top_p
    _, _, pool, probs = E.sampling_choose(logits, S.TOP_P, max_k=8, top_p=0.6, pool_size=50)
    assert pool == [0]                                                  # p0 = 0.636 >= 0.6 closes the pool
    _, _, pool, probs = E.sampling_choose(logits, S.STD, max_k=8, top_p=0.99, pool_size=4)
    assert pool == [0, 1, 2, 3] and abs(sum(probs) - 1.0) < 1e-5         # softmax over the 4-entry pool only
    # the draws follow the pool probabilities
    ids, _, pool, probs = E.sampling_choose(logits, S.TOP_K, max_k=4, top_p=1.0, pool_size=50, seed=7, n_draws=4000)
    freq = np.bincount(ids, minlength=10)[:4] / 4000.0
    assert np.abs(freq - np.array(probs)).max() < 0.03
    with pytest.raises(E.EngineError):
        E.sampling_choose(logits, 7)                                    # not a member of the Std family: use sampling_choose_ex


def test_perplexity_token_nll_is_the_tools_float_log_softmax():
    """perplexity.cc:100-119: float max, double sum of expf(logit - max), nll = -(logit[tok] - max - log(sum))."""
    import ctypes as C
    from inferflow_amd import _capi
    rng = np.random.default_rng(4)
    for vocab in (7, 1000, 32000):
        lg = rng.normal(0, 3.0, vocab).astype(np.float16)
        lg[vocab // 2] = np.float16(9.5)
        for tok in (0, vocab // 2, vocab - 1):
            got = _capi.lib().ifa_perplexity_token_nll(lg.ctypes.data_as(C.c_void_p), vocab, tok)
            f = lg.astype(np.float32)
            m = f.max()
            want = -(float(f[tok] - m) - float(np.log(np.exp(f - m, dtype=np.float32).astype(np.float64).sum())))
            assert abs(got - want) <= 1e-6 * max(1.0, abs(want))
    assert _capi.lib().ifa_perplexity_token_nll(lg.ctypes.data_as(C.c_void_p), 10, 10) < 0
    sub = np.array([0x0001, 0x03FF, 0x8001, 0x7BFF, 0x0000], np.uint16).view(np.float16)        # subnormals, max, zero
    got = _capi.lib().ifa_perplexity_token_nll(sub.ctypes.data_as(C.c_void_p), 5, 3)
    assert abs(got) < 1e-6                                                                        # 65504 dominates: p = 1


@pytest.mark.parametrize("strategy", [S.MIN_P, S.TFS, S.TYPICAL, S.MIROSTAT])
@pytest.mark.parametrize("temperature", [1.0, 0.6])
def test_pool_cutting_strategies_match_the_restated_reference(strategy, temperature):
    """min_p / tfs / typical / mirostat: the same pool and the same draws as oracle/sampling.py, Mirostat's mu carried
    from draw to draw like the query's state."""
    rng = np.random.default_rng(7 * strategy + int(10 * temperature))
    for case, vocab in enumerate([1000, 60, 3, 32000, 2]):
        logits = rng.normal(0, 2.5, vocab).astype(np.float16)
        if case == 1:
            logits[5:9] = np.float16(4.0)
        seed = 99 + case
        r = S.JavaRandom(seed)
        mu_o, mu_e = None, None
        for d in range(12):
            (tok, p), cut, mu_o = S.choose_tokens_ex(logits, strategy, r, temperature=temperature, mu=mu_o)
            # the engine side is stateless per call: replay the generator to draw d by asking for d + 1 draws
            ids, probs, pool_ids, pool_probs, mu_e = E.sampling_choose_ex(logits, strategy, temperature=temperature, seed=seed, n_draws=d + 1)
            assert ids[d] == tok, (case, d)
            assert abs(probs[d] - float(p)) <= 3e-6 * max(1.0, float(p))
            if strategy != S.MIROSTAT:            # (Mirostat's pool depends on mu: compared through the drawn tokens and mu)
                assert pool_ids == [i for i, _ in cut]
                assert np.allclose(pool_probs, [float(w) for _, w in cut], rtol=2e-5, atol=1e-7)
        if strategy == S.MIROSTAT:
            assert abs(mu_e - float(mu_o)) <= 1e-4 * max(1.0, abs(float(mu_o)))


def test_pool_cutting_rules_on_a_known_distribution():
    logits = np.log(np.array([0.5, 0.2, 0.1, 0.08, 0.05, 0.04, 0.02, 0.01], np.float64)).astype(np.float16)
    _, _, pool, probs, _ = E.sampling_choose_ex(logits, S.MIN_P, min_p=0.15)
    assert pool == [0, 1, 2, 3]                                   # p >= 0.15 * 0.5 = 0.075
    _, _, pool, _, _ = E.sampling_choose_ex(logits, S.MIN_P, min_p=0.5)
    assert pool == [0]
    _, _, pool, _, _ = E.sampling_choose_ex(logits, S.TYPICAL, typical_p=0.3)
    assert len(pool) >= 1 and pool[0] in range(8)
    # Mirostat: mu = 2 tau = 2 keeps tokens with surprise <= 2 bits (p >= 0.25): only token 0, so every draw is token 0 ...
    ids, _, pool, _, mu = E.sampling_choose_ex(logits, S.MIROSTAT, tau=1.0, eta=0.5, seed=3, n_draws=3)
    assert ids[0] == 0
    # ... and the observed surprise (0 bits after renormalising a 1-token pool) raises mu by eta * tau each time, admitting more tokens
    ids, _, pool, _, mu = E.sampling_choose_ex(logits, S.MIROSTAT, tau=1.0, eta=0.5, seed=3, n_draws=1)
    assert abs(mu - 2.5) < 1e-5
    with pytest.raises(E.EngineError):
        E.sampling_choose_ex(logits, 11)                          # past the last SamplingStrategyId


@pytest.mark.parametrize("strategy", [S.FSD, S.RANDOM_FSD])
def test_fsd_strategies_match_the_restated_reference(strategy):
    """FSD / RandomizedFSD: top-6 probabilities discounted by the n-gram model of the query's own text (prompt + what was
    selected so far); RandomizedFSD tosses the generator's coin per token for its first 10 tokens."""
    rng = np.random.default_rng(31 + strategy)
    for case in range(4):
        vocab = [50, 200, 12, 32000][case]
        logits = rng.normal(0, 1.5, vocab).astype(np.float16)
        text = [int(t) for t in rng.integers(0, min(vocab, 8), [40, 3, 1, 25][case])]      # a small alphabet: plenty of repeated n-grams
        seed, n = 500 + case, 16
        ids, probs, pool_ids, pool_probs, _ = E.sampling_choose_ex(logits, strategy, seed=seed, n_draws=n, text=text, top_p=0.93)
        r, st = S.JavaRandom(seed), S.FsdState()
        for d in range(n):
            (tok, w), cut = S.choose_tokens_fsd(logits, strategy, r, st, text)
            assert ids[d] == tok, (case, d)
            assert abs(probs[d] - float(w)) <= 3e-6
        assert pool_ids == [i for i, _ in cut]
    # the penalty bites: a candidate that always followed the current context loses against a fresh one of similar probability
    logits = np.full(10, -10.0, np.float16); logits[3] = 2.0; logits[4] = 1.9
    ids, _, pool, w, _ = E.sampling_choose_ex(logits, S.FSD, n_draws=1, text=[1, 2, 3, 1, 2, 3, 1, 2])
    assert ids[0] == 4 and pool[0] == 4 and w[pool.index(3)] < 0.2
