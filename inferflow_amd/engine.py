"""Python view of the C++ InferenceEngine facade (inferflow_amd/host/inference_engine.h) through
include/inferflow_engine.h -- the reference's serving loop, same names and return conventions:

    eng = InferenceEngine.from_ini("llm_inference.ini", "transformer_engine")
    qid = eng.add_query(tokens)              # > 0 id, 0 busy, < 0 error
    while ...:
        for query_id, next_token in eng.infer():
            eng.commit({query_id: (next_token, False)})
    eng.remove_query(qid)
"""
import ctypes as C

import numpy as np

from . import _capi


class EngineError(RuntimeError):
    pass


class InferenceEngine:
    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def _err():
        return _capi.lib().ifa_engine_last_error().decode(errors="replace")

    @classmethod
    def from_ini(cls, ini_path, section="transformer_engine", data_root_dir=""):
        h = _capi.lib().ifa_engine_create(str(ini_path).encode(), section.encode(), data_root_dir.encode())
        if not h:
            raise EngineError("LoadConfig/Init failed: " + cls._err())
        return cls(h)

    def close(self):
        if self._h:
            _capi.lib().ifa_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_query(self, tokens, strategy=None, seed=0, temperature=1.0):
        """strategy: None (the model's default) or a name / SamplingStrategyId ("sample.top_p", "greedy", 1 ...)"""
        arr = (C.c_int * len(tokens))(*[int(t) for t in tokens])
        if strategy is None and seed == 0 and temperature == 1.0:
            return _capi.lib().ifa_engine_add_query(self._h, arr, len(tokens))
        sid = 0 if strategy is None else (strategy if isinstance(strategy, int) else self.strategy_id(strategy))
        return _capi.lib().ifa_engine_add_query_ex(self._h, arr, len(tokens), int(sid), int(seed), float(temperature))

    def strategy_id(self, name=""):
        return _capi.lib().ifa_engine_strategy_id(self._h, (name or "").encode())

    def query_count(self):
        return _capi.lib().ifa_engine_query_count(self._h)

    def remove_query(self, query_id):
        return bool(_capi.lib().ifa_engine_remove_query(self._h, query_id))

    def infer(self, capacity=64):
        """One Infer(): [(query_id, greedy next token)] for the queries that advanced."""
        ids = (C.c_int * capacity)()
        toks = (C.c_int * capacity)()
        n = _capi.lib().ifa_engine_infer(self._h, ids, toks, capacity)
        if n < 0:
            raise EngineError("Infer failed: " + self._err())
        return [(ids[i], toks[i]) for i in range(min(n, capacity))]

    def commit(self, query_map):
        """query_map: {query_id: token} or {query_id: (token, is_end)}"""
        n = len(query_map)
        ids = (C.c_int * n)(); toks = (C.c_int * n)(); ends = (C.c_int * n)()
        for i, (q, v) in enumerate(query_map.items()):
            tok, end = v if isinstance(v, tuple) else (v, False)
            ids[i], toks[i], ends[i] = q, int(tok), int(bool(end))
        return bool(_capi.lib().ifa_engine_commit(self._h, ids, toks, ends, n))

    def perf_stat(self):
        """{key: ms} of the last infer(): key 0 end to end; in study mode the reference's per-phase keys (include/inferflow_engine.h)"""
        keys = (C.c_uint * 256)(); ms = (C.c_float * 256)()
        n = _capi.lib().ifa_engine_perf_stat(self._h, keys, ms, 256)
        if n < 0:
            raise EngineError(self._err())
        return {int(keys[i]): float(ms[i]) for i in range(min(n, 256))}

    def last_logits(self, query_id):
        rows, cols = C.c_int(0), C.c_int(0)
        if not _capi.lib().ifa_engine_last_logits(self._h, query_id, None, 0, C.byref(rows), C.byref(cols)):
            raise EngineError(self._err())
        out = np.zeros((rows.value, cols.value), np.float16)
        if out.size:
            _capi.lib().ifa_engine_last_logits(self._h, query_id, out.ctypes.data_as(C.c_void_p), out.size, C.byref(rows), C.byref(cols))
        return out

    def generate(self, query_id, n_steps):
        out = (C.c_int * max(1, n_steps))()
        ms = C.c_float(0)
        n = _capi.lib().ifa_engine_generate(self._h, query_id, n_steps, out, C.byref(ms))
        if n < 0:
            raise EngineError("Generate failed: " + self._err())
        return [out[i] for i in range(n)], ms.value

    def perplexity(self, tokens, max_length=512, stride=512):
        """(PPL, error estimate, scored tokens) of a token-id stream -- the reference's perplexity tool."""
        arr = (C.c_int * len(tokens))(*[int(t) for t in tokens])
        ppl, err, cnt = C.c_double(0), C.c_double(0), C.c_longlong(0)
        if not _capi.lib().ifa_engine_perplexity(self._h, arr, len(tokens), max_length, stride, C.byref(ppl), C.byref(err), C.byref(cnt)):
            raise EngineError("perplexity failed: " + self._err())
        return ppl.value, err.value, cnt.value

    def model_info(self, key):
        return _capi.lib().ifa_engine_model_info(self._h, key.encode())

    def worker_plan(self, rank):
        """{stage, n_stages, tp_rank, tp_size, layer0, layer1} of partition rank `rank` (None: no such rank)"""
        out = (C.c_int * 6)()
        if _capi.lib().ifa_engine_worker_plan(self._h, int(rank), out) != 0:
            return None
        return dict(zip(("stage", "n_stages", "tp_rank", "tp_size", "layer0", "layer1"), [int(v) for v in out]))

    def worker_tensor(self, rank, layer, tid, expert=-1):
        """(dtype, uint8 host copy, rows, cols) of the slice of tensor (layer local to the rank, tid) rank `rank` holds -- reference
        layout bytes -- or None.  Test surface: the ranks' slices put together are the model the oracle is given."""
        from . import dtypes as dt
        L = _capi.lib()
        h = L.ifa_engine_worker(self._h, int(rank))
        if not h:
            return None
        d, p, r, c = C.c_int(), C.c_void_p(), C.c_size_t(), C.c_size_t()
        if expert >= 0:
            rc = L.ifa_model_get_expert_tensor(C.c_void_p(h), layer, expert, tid, C.byref(d), C.byref(p), C.byref(r), C.byref(c))
        else:
            rc = L.ifa_model_get_tensor(C.c_void_p(h), layer, tid, C.byref(d), C.byref(p), C.byref(r), C.byref(c))
        if rc != 0:
            return None
        nbytes = r.value * dt.row_bytes(d.value, c.value)
        out = np.empty(nbytes, np.uint8)
        _capi.check(L.ifa_memcpy_d2h(out.ctypes.data_as(C.c_void_p), p, nbytes, None))
        _capi.check(L.ifa_stream_sync(None))
        return d.value, out, r.value, c.value


def sampling_choose(logits_f16, strategy_id, max_k=8, top_p=0.9, pool_size=50, temperature=1.0, seed=1, n_draws=1):
    """Host-only StdSamplingStrategy::ChooseTokens on one F16 logits row: (drawn ids, their probabilities, pool ids, pool probs)."""
    lg = np.ascontiguousarray(logits_f16, np.float16)
    ids = (C.c_int * max(1, n_draws))(); pr = (C.c_float * max(1, n_draws))()
    pid = (C.c_int * 256)(); ppr = (C.c_float * 256)()
    n = _capi.lib().ifa_sampling_choose(lg.ctypes.data_as(C.c_void_p), lg.size, int(strategy_id), max_k, top_p, pool_size, temperature,
                                        int(seed), n_draws, ids, pr, pid, ppr, 256)
    if n < 0:
        raise EngineError(_capi.lib().ifa_engine_last_error().decode(errors="replace"))
    return [ids[i] for i in range(n_draws)], [pr[i] for i in range(n_draws)], [pid[i] for i in range(min(n, 256))], [ppr[i] for i in range(min(n, 256))]


def sampling_choose_ex(logits_f16, strategy_id, temperature=1.0, seed=1, n_draws=1, mu=None, max_k=8, top_p=0.9, pool_size=50, min_p=0.05,
                       z=0.95, typical_p=0.95, eta=0.1, tau=5.0, text=()):
    """Any restated strategy (incl. min_p 7, tfs 8, typical 9, mirostat 10): (ids, probs, pool ids, pool probs, mu after the draws)."""
    lg = np.ascontiguousarray(logits_f16, np.float16)
    params = (C.c_float * 9)(max_k, top_p, pool_size, min_p, z, typical_p, eta, tau, 0)
    ids = (C.c_int * max(1, n_draws))(); pr = (C.c_float * max(1, n_draws))()
    pid = (C.c_int * 256)(); ppr = (C.c_float * 256)()
    m = C.c_float(float("nan") if mu is None else mu)
    txt = (C.c_int * max(1, len(text)))(*[int(t) for t in text])
    n = _capi.lib().ifa_sampling_choose_ex(lg.ctypes.data_as(C.c_void_p), lg.size, int(strategy_id), params, temperature, int(seed), n_draws,
                                           ids, pr, pid, ppr, 256, C.byref(m), txt, len(text))
    if n < 0:
        raise EngineError(_capi.lib().ifa_engine_last_error().decode(errors="replace"))
    return [ids[i] for i in range(n_draws)], [pr[i] for i in range(n_draws)], [pid[i] for i in range(min(n, 256))], [ppr[i] for i in range(min(n, 256))], m.value


def random_doubles(seed, n):
    out = (C.c_double * n)()
    _capi.lib().ifa_sampling_random_doubles(int(seed), n, out)
    return [out[i] for i in range(n)]
