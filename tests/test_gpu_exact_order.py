"""-m gpu: option `exact_order` (csrc/ifa_exact.hip, csrc/ifa_engine_exact.hip) -- single-token steps in the summation order of the
reference's CUDA kernels as the oracle restates them.  Everything a step leaves behind must be BIT-IDENTICAL to oracle.Model's:
logits, greedy id, the last layer's output, the K / V rows of every layer (F16 rows, or int8 codes + scales of a Q8_B32T2 cache).
Small shapes here, over the weight formats, wirings and cache formats the step accepts; the headline configurations at their own
size are in tests/test_gpu_fullsize_oracle.py."""
import numpy as np
import pytest

from inferflow_amd import dtypes as dt
from tests.test_gpu_engine import _build_custom, MINICPM_LIKE

pytestmark = pytest.mark.gpu

CASES = [
    ("q4_f16kv", "test_gqa", dt.Q4_B32T1A, dt.F16, dict(), False),
    ("q4b_q8kv_bias", "test_gqa", dt.Q4_B32T1B, dt.Q8_B32T2, dict(), True),
    ("q3h_q8kv", "test_gqa", dt.Q3H_B64T1, dt.Q8_B32T2, dict(), False),
    ("q4b64", "test_mha", dt.Q4_B64T1, dt.F16, dict(), False),
    ("q5b64_rope1", "test_gqa", dt.Q5_B64T1, dt.F16, dict(rope_order=1), False),
    ("q6b64", "test_mha", dt.Q6_B64T1, dt.Q8_B32T2, dict(), True),
    ("q8t2_relu", "test_gqa", dt.Q8_B32T2, dt.F16, dict(act_kind=2), False),
    ("f16_weights", "test_mha", dt.F16, dt.F16, dict(), True),
    ("q5b32_f16_activations", "test_gqa", dt.Q5_B32T1, dt.F16, dict(), False),       # not on the int8 path: dequantised weights x F16 row
    ("q2_no_glu", "test_gqa", dt.Q2_B32T1A, dt.Q8_B32T2, dict(is_glu=0), False),
    ("minicpm_scales_norm_bases", "test_gqa", dt.Q4_B32T1A, dt.F16, dict(MINICPM_LIKE, attn_norm_base=1.0, ffn_norm_base=1.0, out_norm_base=1.0), False),
]


def _same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8).reshape(-1), np.ascontiguousarray(b).view(np.uint8).reshape(-1))


@pytest.mark.parametrize("name,shape,wd,kvd,cfg,bias", CASES, ids=[c[0] for c in CASES])
def test_exact_order_steps_are_bit_identical_to_the_oracle(name, shape, wd, kvd, cfg, bias):
    max_ctx = 48
    wk, om, s = _build_custom(shape, wd, kvd, max_ctx, cfg, with_bias=bias)
    wk.set_option("exact_order", 1)
    om.capture_layers(True)
    rng = np.random.default_rng(5)
    prompt = rng.integers(0, s["vocab"], 5).astype(np.int32)
    # the prompt through ifa_model_forward: in this mode its rows go one by one through the single-row step, like the oracle's here
    tok = wk.forward(prompt, 0)
    for i, t in enumerate(prompt):
        tok_o, lg_o = om.forward(np.array([t], np.int32), i)
    assert _same_bits(wk.read_buffer("logits").view(np.uint16), lg_o[0].view(np.uint16)), "prompt logits"
    assert int(tok) == int(tok_o)
    pos = len(prompt)
    cur = int(tok_o)
    for step in range(30):            # free-running on both sides: identical logits -> identical ids, contexts up to 35 keys
        toks, _ = wk.decode(cur, pos, 1)
        tok_o, lg_o = om.forward(np.array([cur], np.int32), pos)
        assert _same_bits(wk.read_buffer("logits").view(np.uint16), lg_o[0].view(np.uint16)), "logits of step %d" % step
        assert int(toks[0]) == int(tok_o), "greedy id of step %d" % step
        if "out_scale" not in cfg:        # (TensorOpr::Scale of the output runs in place on both sides, after the oracle's tap)
            assert _same_bits(wk.read_buffer("x").view(np.uint16), om.layer_io()[s["layers"]].view(np.uint16)), "last layer's output, step %d" % step
        cur, pos = int(tok_o), pos + 1
    for l in range(s["layers"]):      # every layer's cache rows: F16 rows or int8 codes + scales
        rb = om.kv_rows(l, 0, pos).shape[1]
        assert _same_bits(wk.read_buffer("kcache", layer=l, nbytes=pos * rb), om.kv_rows(l, 0, pos)), "K rows of layer %d" % l
        assert _same_bits(wk.read_buffer("vcache", layer=l, nbytes=pos * rb), om.kv_rows(l, 1, pos)), "V rows of layer %d" % l
    # several steps per call take the same route
    wk.set_option("exact_order", 1)
    toks, _ = wk.decode(cur, pos, 3)
    for t in toks:
        tok_o, _ = om.forward(np.array([cur], np.int32), pos)
        assert int(t) == int(tok_o)
        cur, pos = int(tok_o), pos + 1
    wk.close()


def test_exact_order_refuses_models_outside_the_step():
    """GELU (tanhf) and Std-norm models have no order-exact step: the call fails, it never runs other arithmetic silently."""
    from tests.test_gpu_engine import BLOOM_LIKE
    wk, om, s = _build_custom("test_gqa", dt.Q4_B32T1A, dt.F16, 32, BLOOM_LIKE, with_bias=True)
    wk.set_option("exact_order", 1)
    with pytest.raises(Exception, match="exact_order"):
        wk.decode(3, 0, 1)
    wk.set_option("exact_order", 0)
    wk.decode(3, 0, 1)
    wk.close()


@pytest.mark.parametrize("shape,wd,kvd,layers,steps", [("test_moe", dt.Q4_B32T1A, dt.F16, None, 30), ("test_moe", dt.Q3H_B64T1, dt.Q8_B32T2, None, 30),
                                                     ("mixtral_8x7b", dt.Q4_B32T1A, dt.F16, 4, 10), ("mixtral_8x7b", dt.Q4_B32T1A, dt.F16, 32, 12)],
                         ids=["small_q4", "small_q3h_kvq8", "mixtral_widths_4_layers", "mixtral_8x7b_all_32_layers"])
def test_exact_order_mixture_of_experts_is_bit_identical_to_the_oracle(shape, wd, kvd, layers, steps):
    """Mixture-of-experts layers in the order-exact step: F16 router GEMV, 32-lane softmax, top-k by BuildRowsForMoE's rules, the
    selected experts' FFNs in ascending order, out x w + f accumulated as the oracle restates AddByRowIdx.  With both sides in ONE
    summation order a router near tie is no longer a discontinuity between them (rounds 5's finding: at 32 random-init layers 55 of
    56 rows ran different experts on the two sides): logits, ids and K / V rows are the oracle's bit for bit, free-running."""
    import oracle as o
    from inferflow_amd import synth
    from tests.model_util import oracle_model_from_host
    max_ctx = 48
    kw = dict(layers=layers) if layers else {}
    wk, host, s = synth.build(shape, wd, kvd, max_ctx=max_ctx, quant_threshold=0, std=0.06 if shape == "test_moe" else 0.02, keep_host=shape == "test_moe", **kw)
    if shape == "test_moe":
        om = oracle_model_from_host(host, s, max_ctx, kvd)
    else:      # full widths: the oracle gets the blocks the device quantiser wrote, read back (tests/test_gpu_fullsize_oracle.py does the same)
        om = o.Model(dim=s["dim"], layers=s["layers"], heads=s["heads"], kv_heads=s["kv_heads"], head_dim=s["head_dim"], ffn=s["ffn"], vocab=s["vocab"],
                     max_ctx=max_ctx, kv_dtype=kvd, experts=s["experts"], moe_top_k=s["moe_top_k"])
        for key in [(-1, t) for t in (0, 1, 3)] + [(l, t) for l in range(s["layers"]) for t in (10, 12, 13, 14, 15, 16, 21)]:
            d, data, rows, cols = wk.get_tensor_host(max(key[0], 0), key[1])
            om.set_tensor(max(key[0], 0), key[1], d, data.reshape(rows, cols) if d == dt.F16 else data.reshape(rows, -1), rows, cols)
        for l in range(s["layers"]):
            for e in range(s["experts"]):
                for t in (18, 19, 20):
                    d, data, rows, cols = wk.get_expert_tensor_host(l, e, t)
                    om.set_tensor(l, t, d, data.reshape(rows, -1), rows, cols, expert=e)
    wk.set_option("exact_order", 1)
    prompt = np.random.default_rng(9).integers(3, s["vocab"], 4).astype(np.int32)
    cur, margins = None, []
    for i in range(len(prompt) + steps):
        tok_in = int(prompt[i]) if i < len(prompt) else cur
        toks, _ = wk.decode(tok_in, i, 1)
        tok_o, lg_o = om.forward(np.array([tok_in], np.int32), i)
        margins.append(om.moe_margin())
        assert _same_bits(wk.read_buffer("logits").view(np.uint16), lg_o[0].view(np.uint16)), "logits of step %d" % i
        assert int(toks[0]) == int(tok_o), "greedy id of step %d" % i
        cur = int(tok_o)
    n = len(prompt) + steps
    for l in range(s["layers"]):
        rb = om.kv_rows(l, 0, n).shape[1]
        assert _same_bits(wk.read_buffer("kcache", layer=l, nbytes=n * rb), om.kv_rows(l, 0, n))
        assert _same_bits(wk.read_buffer("vcache", layer=l, nbytes=n * rb), om.kv_rows(l, 1, n))
    print("order-exact MoE %s: %d steps bit-identical; smallest router margin met %.5f" % (shape, n, min(margins)))
    wk.close()


def test_exact_order_batched_step_runs_every_query_through_the_exact_row():
    """ifa_model_decode_batch with exact_order on: three queries on their own KV cache sets, each row through the single-row step --
    every query's logits and ids are those of an oracle of its own, bit for bit (a batched step of the reference is n independent rows)."""
    import torch
    wk, om0, s = _build_custom("test_gqa", dt.Q4_B32T1A, dt.Q8_B32T2, 40, dict(), with_bias=False)
    oms = [om0] + [_build_custom("test_gqa", dt.Q4_B32T1A, dt.Q8_B32T2, 40, dict(), with_bias=False)[1] for _ in range(2)]      # (same seed: same weights)
    wk.kv_slots(3)
    wk.set_option("exact_order", 1)
    rng = np.random.default_rng(3)
    cur = [int(t) for t in rng.integers(0, s["vocab"], 3)]
    pos = [0, 0, 0]
    lg = torch.empty((3, s["vocab"]), dtype=torch.float16, device="cuda")
    for step in range(12):
        act = [0, 1, 2] if step % 3 else [2, 0]          # (ragged: not every query advances in every step)
        nxt = wk.decode_batch([cur[q] for q in act], [pos[q] for q in act], act, lg)
        rows = lg.cpu().numpy().view(np.uint16)
        for j, q in enumerate(act):
            tok_o, lg_o = oms[q].forward(np.array([cur[q]], np.int32), pos[q])
            assert _same_bits(rows[j], lg_o[0].view(np.uint16)), "query %d step %d" % (q, step)
            assert int(nxt[j]) == int(tok_o)
            cur[q], pos[q] = int(tok_o), pos[q] + 1
    wk.close()
