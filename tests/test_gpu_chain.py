"""The chained FFN launch (csrc/ifa_decode_chain.h, option fuse_ffn): [Wo ->] W1 | W3 -> W2 of a layer as ONE launch whose next
rows are requested before the hand-off.  Tokens, last-step logits and the KV cache must be bit-identical to the separate
launches (same row / prologue / epilogue code, the hand-off is the only difference), and the launch must be the one that runs."""
import numpy as np
import pytest

from inferflow_amd import dtypes as dt, synth

pytestmark = pytest.mark.gpu

CASES = [
    ("llama2_7b", dt.Q4_B32T1A, dt.F16, 4, {}),
    ("llama2_7b", dt.Q3H_B64T1, dt.Q8_B32T2, 4, {}),
    ("llama2_7b", dt.Q4_B32T1A, dt.F16, 3, {"kv_heads": 8, "ffn": 14336}),       # Mixtral's dense shape: two passes of W1 | W3 rows
]


def _run(wk, s, prompt, steps, **opts):
    for k, v in opts.items():
        wk.set_option(k, v)
    wk.reset()
    tok = wk.forward(prompt, 0)
    toks, _ = wk.decode(int(tok), len(prompt), steps)
    logits = wk.read_buffer("logits").view(np.uint16).copy()
    kc = wk.read_buffer("kcache", layer=s["layers"] - 1).copy()
    vc = wk.read_buffer("vcache", layer=s["layers"] - 1).copy()
    return list(toks), logits, kc, vc


@pytest.mark.parametrize("shape,wd,kvd,layers,kw", CASES, ids=["llama7b_q4_f16", "llama7b_q3h_kvq8", "gqa8_ffn14336_q4"])
def test_chained_ffn_launch_is_bit_identical(shape, wd, kvd, layers, kw):
    wk, _, s = synth.build(shape, wd, kvd, max_ctx=320, layers=layers, **kw)
    prompt = (np.arange(20, dtype=np.int32) * 11 + 5) % s["vocab"]
    for steps in (40, 250):      # (positions past 255: the tag's position bits wrap)
        ref = _run(wk, s, prompt, steps, fuse_ffn=0)
        for mode in (1, 2):
            got = _run(wk, s, prompt, steps, fuse_ffn=mode)
            assert got[0] == ref[0], "tokens differ (fuse_ffn %d, %d steps)" % (mode, steps)
            assert np.array_equal(got[1], ref[1]), "logits differ (fuse_ffn %d, %d steps)" % (mode, steps)
            assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), "KV cache differs (fuse_ffn %d, %d steps)" % (mode, steps)
            # the chained launch is really the one that ran: its timing entry point refuses models it does not take
            assert wk.time_kernel(9, 4) > 0.0
    # several calls in a row (the call counter in the tag), with and without the fused attention in front
    wk.set_option("fuse_ffn", 2)
    a = _run(wk, s, prompt, 9, fuse_attn=0)
    b = _run(wk, s, prompt, 9, fuse_attn=1)
    c = _run(wk, s, prompt, 9, fuse_ffn=0)
    assert a[0] == b[0] == c[0] and np.array_equal(a[1], c[1]) and np.array_equal(b[1], c[1])


def test_chained_launch_declines_shapes_it_has_no_kernel_for():
    wk, _, s = synth.build("tiny15m", dt.F16, dt.F16, max_ctx=128)
    prompt = np.arange(3, 11, dtype=np.int32)
    a = _run(wk, s, prompt, 24, fuse_ffn=2)
    b = _run(wk, s, prompt, 24, fuse_ffn=0)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    with pytest.raises(Exception):
        wk.time_kernel(9, 2)


def test_q3h_native_32_byte_stream_is_bit_identical_to_the_nibble_pair_stream():
    """Option q3h_native: Wo / W1 / W3 / W2 of a Q3H_B64T1 model streamed at the format's own 32 bytes per 64
    weights (pair codes decoded in the kernel: csrc/ifa_decode_formats.h WRowQ3HN) instead of the 36-byte nibble pairs: the same
    integer dots, so tokens, logits and KV rows must not change by a bit -- at Llama-2-7B widths (4096 / 11008 columns: one and
    three blocks per lane) and at a small GQA shape."""
    for shape, kw in (("llama2_7b", dict(layers=3)), ("test_gqa", dict())):
        wk, _, s = synth.build(shape, dt.Q3H_B64T1, dt.Q8_B32T2, max_ctx=160, **kw)
        prompt = (np.arange(9, dtype=np.int32) * 7 + 3) % s["vocab"]

        def run(native):
            wk.set_option("q3h_native", native)
            wk.reset()
            tok = wk.forward(prompt, 0)
            toks, _ = wk.decode(int(tok), len(prompt), 60)
            return (list(toks), wk.read_buffer("logits").view(np.uint16).copy(), wk.read_buffer("kcache", layer=s["layers"] - 1).copy())

        a, b = run(0), run(1)
        assert a[0] == b[0], shape
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), shape
        wk.set_option("q3h_native", 0)
        wk.close()
