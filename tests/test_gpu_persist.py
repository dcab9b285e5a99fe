"""Persistent decode launch (csrc/ifa_decode_persist.h, option "persist") against the five-launch fused step: the same
greedy tokens and bit-identical logits, every hand-off of every layer bit-identical to the buffer the five-launch path
leaves (the five-launch path itself is held to the oracle in test_gpu_engine.py / test_gpu_ref_model.py).
Reference sequence: src/transformer/inference_worker.cc:762-981, :1116-1312."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not __import__("inferflow_amd").lib().ifa_experimental_built(),
                                 reason="the persistent launch is a parked dead end (csrc/experimental/): built with IFA_EXPERIMENTAL=1 only")]

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from inferflow_amd import dtypes as dt, synth, worker as W  # noqa: E402


def _build(shape, wdtype, kv_dtype, max_ctx=1024, **over):
    wk, _, s = synth.build(shape, wdtype, kv_dtype, max_ctx=max_ctx, quant_threshold=0, **over)
    dev = "cuda:0"
    for layer in range(s["layers"]):      # non-trivial norm weights (synth uses ones)
        for tid, seed in ((W.T_ATTN_NORM, 7000), (W.T_FFN_NORM, 8000)):
            w = (1.0 + synth.gen_f16((s["dim"],), seed + layer, 0.1, dev).float()).half()
            wk.set_tensor_f16(layer, tid, dt.F16, w, 1, s["dim"])
    return wk, s


def _step(wk, tok, pos, persist, steps, layers=0):
    wk.set_option("persist", persist)
    wk.set_option("debug_layers", layers)
    # layers > 0: the layer outputs are read back from x / x2 -- keep the step's last launch from gathering the NEXT step's
    # embedding row into x (step_tail, csrc/ifa_decode_lmhead_tail.h); the full steps run with it
    wk.set_option("step_tail", 0 if layers else 1)
    toks, _ = wk.decode(tok, pos, steps)
    return toks, wk.read_buffer("logits").view(np.uint16).copy()


def _arena(wk, s):
    raw = wk.read_buffer("ps_arena").view(np.uint64)
    QD, KVD = s["heads"] * s["head_dim"], s["kv_heads"] * s["head_dim"]
    counts = [s["dim"] // 2, (QD + 2 * KVD) // 2, QD // 4 + QD // 16, s["dim"] // 2, s["ffn"] // 2]
    out, off = {}, 0
    for name, n in zip(("x", "qkv", "att", "a", "act"), counts):
        g = raw[off:off + n]
        out[name] = ((g >> np.uint64(32)).astype(np.uint32), (g & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        off += (n + 63) // 64 * 64
    return out


CASES = [("test_gqa", dt.Q4_B32T1A, dt.F16, {}), ("test_gqa", dt.Q4_B32T1A, dt.Q8_B32T2, {}), ("test_gqa", dt.Q3H_B64T1, dt.F16, {}),
         ("test_gqa", dt.Q3H_B64T1, dt.Q8_B32T2, {}), ("test_gqa", dt.Q4_B32T1A, dt.F16, dict(rope_order=1))]


@pytest.mark.parametrize("shape,wd,kv,over", CASES)
def test_persistent_launch_matches_five_launch_step(shape, wd, kv, over):
    wk, s = _build(shape, wd, kv, **over)
    prompt = (np.arange(19, dtype=np.int32) * 7 + 3) % s["vocab"]
    tok = int(wk.forward(prompt, 0))
    t_ref, l_ref = _step(wk, tok, len(prompt), 0, 12)
    t_ps, l_ps = _step(wk, tok, len(prompt), 1, 12)
    assert np.array_equal(t_ref, t_ps)
    assert np.array_equal(l_ref, l_ps)          # F16 logits, bit for bit
    # every hand-off of every layer (the step truncated to its first n layers in both paths)
    QD = s["heads"] * s["head_dim"]
    for n in range(1, s["layers"] + 1):
        _step(wk, tok, len(prompt), 0, 1, n)
        ref = dict(qkv=wk.read_buffer("dqkv").view(np.uint16).copy(), attq=wk.read_buffer("attq").copy(),
                   a=wk.read_buffer("a").view(np.uint16).copy(), act=wk.read_buffer("t1").view(np.uint16).copy()[:s["ffn"]],
                   xo=wk.read_buffer("x2" if n % 2 == 1 else "x").view(np.uint16).copy())
        _step(wk, tok, len(prompt), 1, 1, n)
        ar = _arena(wk, s)
        ep = lambda edge: (n - 1) * 8 + edge + 1  # noqa: E731
        for name, edge in (("qkv", 1), ("a", 3), ("act", 4)):
            tags, vals = ar[name]
            assert (tags == ep(edge)).all(), (n, name)
            assert np.array_equal(vals.view(np.uint16), ref[name]), (n, name)
        tags, vals = ar["att"]
        sc = (QD + 15) // 16 * 16
        img = np.concatenate([ref["attq"][:QD].view(np.uint32), ref["attq"][sc:sc + QD // 32 * 8].view(np.uint32)])
        assert (tags == ep(2)).all() and np.array_equal(vals, img), (n, "att")
        assert np.array_equal(wk.read_buffer("x2").view(np.uint16), ref["xo"]), (n, "x_out")
    wk.close()


def test_persistent_launch_llama2_7b_width_two_layers():
    """Llama-2-7B's widths (4096 / 11008, 32 heads of 128), 2 layers: the shape the bench runs."""
    wk, s = _build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=512, layers=2, vocab=4096)
    prompt = (np.arange(33, dtype=np.int32) * 5 + 1) % s["vocab"]
    tok = int(wk.forward(prompt, 0))
    t_ref, l_ref = _step(wk, tok, len(prompt), 0, 24)
    t_ps, l_ps = _step(wk, tok, len(prompt), 1, 24)
    assert np.array_equal(t_ref, t_ps) and np.array_equal(l_ref, l_ps)
    wk.close()


def test_persistent_launch_falls_back_for_unsupported_models():
    """MoE / Std-norm / small-head models keep the five-launch step with persist=1 (same tokens as persist=0)."""
    wk, _, s = synth.build("test_mha", dt.Q4_B32T1A, dt.F16, max_ctx=256, quant_threshold=0)      # head_dim 32: no persistent kernel
    prompt = (np.arange(9, dtype=np.int32) * 3 + 2) % s["vocab"]
    tok = int(wk.forward(prompt, 0))
    a, _ = _step(wk, tok, len(prompt), 0, 6)
    b, _ = _step(wk, tok, len(prompt), 1, 6)
    assert np.array_equal(a, b)
    wk.close()
