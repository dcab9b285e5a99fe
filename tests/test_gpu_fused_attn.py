"""The attention as the tail of the QKV launch (csrc/ifa_decode_qkv_attn.h, option fuse_attn): tokens, last-step logits and the KV
cache must be bit-identical to the five-launch step (same kernel bodies, the hand-off is the only difference), and the launch must be
the one that runs for the headline shape.  (The chained FFN launch has its own file: tests/test_gpu_chain.py.)"""
import numpy as np
import pytest

from inferflow_amd import dtypes as dt, synth, worker as W

pytestmark = pytest.mark.gpu

CASES = [
    ("llama2_7b", dt.Q4_B32T1A, dt.F16, 4),        # headline widths (4 layers keep the test fast), 8 workgroups per head
    ("llama2_7b", dt.Q3H_B64T1, dt.Q8_B32T2, 4),   # configs[2]: 3.5-bit weights, 8-bit KV cache
    ("mixtral_dense_like", dt.Q4_B32T1A, dt.F16, 3),
]


def _build(shape, wd, kvd, layers, max_ctx=480):
    if shape == "mixtral_dense_like":       # grouped-query geometry of Mixtral (32 heads, 8 kv heads, dim 4096) with a dense FFN
        return synth.build("llama2_7b", wd, kvd, max_ctx=max_ctx, layers=layers, kv_heads=8, ffn=14336)
    return synth.build(shape, wd, kvd, max_ctx=max_ctx, layers=layers)


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


@pytest.mark.parametrize("shape,wd,kvd,layers", CASES, ids=["llama7b_q4_f16", "llama7b_q3h_kvq8", "gqa8_q4_f16"])
def test_fused_qkv_attention_launch_is_bit_identical(shape, wd, kvd, layers):
    wk, _, s = _build(shape, wd, kvd, layers)
    prompt = (np.arange(20, dtype=np.int32) * 11 + 5) % s["vocab"]
    # contexts that cross the 64 / 128 / 256 prefetch buckets: 20 .. 20 + 250; 20 + 440: far past the 256-row bucket, still one
    # workgroup per head (the default split threshold of these shapes is 512 / 640 keys)
    for steps in (40, 120, 250, 440):
        # (attn_split_ctx 0: one workgroup per head in every run -- the default threshold depends on the options compared here)
        ref = _run(wk, s, prompt, steps, fuse_attn=0, step_tail=0, attn_split_ctx=0)
        wk.set_option("step_tail", 1)
        # attn_unload (default 1): in the 256-row bucket the heads' workgroups take no weight rows (the UL kernels) -- both mappings
        for opts in [{"fuse_attn": 1, "attn_unload": 0}, {"fuse_attn": 1, "attn_unload": 1}]:
            got = _run(wk, s, prompt, steps, **opts)
            assert got[0] == ref[0], "tokens differ (%r, %d steps)" % (opts, steps)
            assert np.array_equal(got[1], ref[1]), "logits differ (%r, %d steps)" % (opts, steps)
            assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), "KV cache differs (%r, %d steps)" % (opts, steps)
    # the fused launch is really the one that ran: its timing entry point refuses models it does not take
    wk.set_option("fuse_attn", 1)
    assert wk.time_kernel(7, 4) > 0.0


def test_fused_launch_declines_shapes_it_has_no_kernel_for():
    # 6 x 48 heads (stories15M): no instance -> the five-launch step runs, same tokens as with the option off
    wk, _, s = synth.build("tiny15m", dt.F16, dt.F16, max_ctx=128)
    prompt = np.arange(3, 11, dtype=np.int32)
    a = _run(wk, s, prompt, 24, fuse_attn=1)
    b = _run(wk, s, prompt, 24, fuse_attn=0)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    with pytest.raises(Exception):
        wk.time_kernel(7, 2)


@pytest.mark.parametrize("shape,wd,kvd", [("llama2_7b", dt.Q4_B32T1A, dt.F16), ("tiny15m", dt.F16, dt.F16), ("tiny15m", dt.Q8_B32T2, dt.F16)],
                         ids=["llama7b_q4", "tiny15m_f16", "tiny15m_q8"])
def test_step_tail_launch_is_bit_identical(shape, wd, kvd):
    """lm_head + argmax + state advance + the next step's gather as ONE launch (csrc/ifa_decode_lmhead_tail.h, option step_tail)
    against the three launches it replaces: tokens and logits of every call, with excluded ids, across several calls and a
    changed workgroup count (rpw_lm) between them."""
    kw = {"layers": 2} if shape == "llama2_7b" else {}
    wk, _, s = synth.build(shape, wd, kvd, max_ctx=256, **kw)
    prompt = (np.arange(9, dtype=np.int32) * 13 + 2) % s["vocab"]

    def run(tail, excl):
        wk.set_option("step_tail", tail)
        wk.set_excluded_tokens(excl)
        out = []
        wk.reset()
        tok = int(wk.forward(prompt, 0))
        pos = len(prompt)
        for rpw, steps in ((0, 17), (1, 5), (3, 30), (0, 1), (0, 40)):
            wk.set_option("rpw_lm", rpw)
            toks, _ = wk.decode(tok, pos, steps)
            out.append((list(toks), wk.read_buffer("logits").view(np.uint16).copy()))
            tok, pos = int(toks[-1]), pos + steps
        wk.set_option("rpw_lm", 0)
        return out

    free = run(0, [])
    first = free[0][0][0]
    for excl in ([], [first], [first, free[0][0][1], 0]):
        a, b = run(0, excl), run(1, excl)
        for (ta, la), (tb, lb) in zip(a, b):
            assert ta == tb, "tokens differ (excluded %r)" % (excl,)
            assert np.array_equal(la, lb), "logits differ (excluded %r)" % (excl,)
        if excl:
            assert all(t not in excl for ts, _ in b for t in ts)
    wk.set_excluded_tokens([])


def test_decode_prepare_runs_no_step_and_changes_no_result():
    """ifa_model_decode_prepare captures what a decode call replays without running a step: the KV cache rows and the tokens /
    logits of the call that follows are those of a worker that was never prepared; graph_steps > 1 (several steps per replay,
    opt-in) gives the same tokens and logits as single-step replays."""
    wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=256, layers=2)
    prompt = (np.arange(12, dtype=np.int32) * 5 + 1) % s["vocab"]
    ref = _run(wk, s, prompt, 21)
    wk.reset()
    tok = wk.forward(prompt, 0)
    kc0 = wk.read_buffer("kcache", layer=1).copy()
    wk.decode_prepare(len(prompt), 21)
    assert np.array_equal(wk.read_buffer("kcache", layer=1), kc0)
    toks, _ = wk.decode(int(tok), len(prompt), 21)
    assert list(toks) == ref[0] and np.array_equal(wk.read_buffer("logits").view(np.uint16), ref[1])
    got = _run(wk, s, prompt, 21, graph_steps=4)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    wk.set_option("graph_steps", 1)
