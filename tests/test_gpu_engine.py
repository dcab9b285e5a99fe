"""-m gpu: the decode worker (op-by-op forward and the fused/graph decode)
against the whole-model oracle on small synthetic models."""
import numpy as np
import pytest
import torch

import oracle as o
from inferflow_amd import dtypes as dt, synth
from tests import gpu_util as g
from tests.model_util import oracle_model_from_host

pytestmark = pytest.mark.gpu

CASES = [
    ("test_gqa", dt.Q4_B32T1A, dt.F16),
    ("test_mha", dt.Q4_B32T1A, dt.Q8_B32T2),
    ("test_gqa", dt.Q4_B32T1B, dt.Q8_B32T2),
]


def _logits_close(a, b):
    a = a.astype(np.float32); b = b.astype(np.float32)
    cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    return cos, float(np.abs(a - b).max())


@pytest.mark.parametrize("shape,wd,kvd", CASES, ids=["gqa_q4_kvf16", "mha_q4_kvq8", "gqa_q4b_kvq8"])
def test_forward_and_fused_decode_match_oracle(shape, wd, kvd):
    max_ctx = 64
    wk, host, s = synth.build(shape, wd, kvd, max_ctx=max_ctx, quant_threshold=0, std=0.06, keep_host=True)
    om = oracle_model_from_host(host, s, max_ctx, kvd)
    ok, why = wk.fused_supported()
    assert ok, why
    rng = np.random.default_rng(42)
    prompt = rng.integers(3, s["vocab"], 9).astype(np.int32)

    # --- prefill (op-by-op, T>1) with full logits, like return_output_tensors
    lg = torch.empty((len(prompt), s["vocab"]), dtype=torch.float16, device="cuda")
    tok_gpu = wk.forward(prompt, 0, lg)
    tok_orc, lg_orc = om.forward(prompt, 0)
    cos, mad = _logits_close(g.host(lg), lg_orc)
    # fp tolerance: device expf/powf/sinf + attention summation order, amplified by
    # re-quantisation; stated bound: cosine >= 0.9995 and |dlogit| <= 0.03 (logit std ~0.5)
    assert cos >= 0.9995 and mad <= 0.03, (cos, mad)
    top2 = np.sort(lg_orc[-1].astype(np.float32))[-2:]
    if top2[1] - top2[0] > 0.05:
        assert tok_gpu == tok_orc

    # --- fused + graph decode vs oracle, greedy, token by token
    n_steps = 12
    toks_fused, ms = wk.decode(tok_gpu, len(prompt), n_steps)
    assert ms > 0
    cur = tok_gpu
    for i in range(n_steps):
        t_or, l_or = om.forward(np.array([cur], np.int32), len(prompt) + i)
        top2 = np.sort(l_or[0].astype(np.float32))[-2:]
        if top2[1] - top2[0] <= 0.05:       # near tie: either choice is within tolerance; follow the GPU
            cur = int(toks_fused[i]); continue
        assert int(toks_fused[i]) == t_or, "step %d" % i
        cur = t_or

    # --- fused decode == op-by-op decode on the same worker (same rounding points)
    wk.reset()
    wk.forward(prompt, 0)
    wk.set_option("fused", 0)
    toks_ops, _ = wk.decode(tok_gpu, len(prompt), n_steps)
    wk.set_option("fused", 1)
    assert np.array_equal(toks_ops, toks_fused)
    # --- and eager fused (no graph) == graph replay, bit for bit
    wk.reset()
    wk.forward(prompt, 0)
    wk.set_option("graph", 0)
    toks_eager, _ = wk.decode(tok_gpu, len(prompt), n_steps)
    lg_eager = wk.read_buffer("logits").copy()
    wk.set_option("graph", 1)
    wk.reset()
    wk.forward(prompt, 0)
    toks_graph, _ = wk.decode(tok_gpu, len(prompt), n_steps)
    assert np.array_equal(toks_eager, toks_graph)
    assert np.array_equal(lg_eager, wk.read_buffer("logits"))
    wk.close()


def test_kv_cache_rows_match_oracle_after_prefill():
    wk, host, s = synth.build("test_gqa", dt.Q4_B32T1A, dt.Q8_B32T2, max_ctx=32, quant_threshold=0, std=0.06, keep_host=True)
    prompt = np.arange(5, 12, dtype=np.int32)
    wk.forward(prompt, 0)
    kv_dim = s["kv_heads"] * s["head_dim"]
    rb = dt.row_bytes(dt.Q8_B32T2, kv_dim)
    k0 = wk.read_buffer("kcache", 0, nbytes=rb * len(prompt)).reshape(len(prompt), rb)
    # layer-0 K rows depend only on embeddings, norm, wk GEMV (fp16-activation path), RoPE, Q8 quantiser
    om = oracle_model_from_host(host, s, 32, dt.Q8_B32T2)
    om.forward(prompt, 0)
    deq_gpu = o.dequantize(dt.Q8_B32T2, k0, kv_dim).astype(np.float32)
    # oracle cache is private; recompute layer-0 K through the op-level oracle instead
    emb = host[(-1, 0)][1].reshape(s["vocab"], s["dim"])[prompt]
    xn = o.rmsnorm(emb, host[(0, 10)][1].reshape(-1))
    wkq = o.quantize(dt.Q4_B32T1A, host[(0, 13)][1].reshape(kv_dim, s["dim"]))
    k = np.stack([o.gemv_f16x(dt.Q4_B32T1A, wkq, kv_dim, s["dim"], xn[t]) for t in range(len(prompt))])
    k = o.rope(k.reshape(len(prompt), s["kv_heads"], s["head_dim"]), 0).reshape(len(prompt), kv_dim)
    deq_orc = o.dequantize(dt.Q8_B32T2, o.quantize_act_q8(k), kv_dim).astype(np.float32)
    assert np.abs(deq_gpu - deq_orc).max() <= 0.02 * np.abs(deq_orc).max()
    wk.close()


def test_decode_rejects_out_of_range_positions():
    wk, _, s = synth.build("test_gqa", max_ctx=16, quant_threshold=0)
    rc = g.capi().ifa_model_decode(wk._h, 1, 10, 10, None, None)
    assert rc == -1 and b"max_ctx" in g.capi().ifa_last_error()
    rc = g.capi().ifa_model_forward(wk._h, None, 1, 0, None, None)
    assert rc == -1
    wk.close()
