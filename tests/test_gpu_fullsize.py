"""-m gpu: BASELINE full-size checks (Llama-2-7B shapes, Q4_B32T1A) through
size-independent properties: fused == unfused bit for bit, decode is deterministic
and idempotent w.r.t. the KV cache, the stream of tokens is identical with and
without graph replay."""
import numpy as np
import pytest

from inferflow_amd import dtypes as dt, synth

pytestmark = pytest.mark.gpu


def test_llama2_7b_fused_equals_unfused_and_is_idempotent():
    wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=64)
    ok, why = wk.fused_supported()
    assert ok, why
    prompt = np.random.default_rng(42).integers(3, s["vocab"], 8).astype(np.int32)
    tok = wk.forward(prompt, 0)
    fused, _ = wk.decode(tok, len(prompt), 6)
    logits_fused = wk.read_buffer("logits").copy()
    # same positions again: KV rows are overwritten with identical values -> identical tokens / logits
    again, _ = wk.decode(tok, len(prompt), 6)
    assert np.array_equal(fused, again)
    assert np.array_equal(logits_fused, wk.read_buffer("logits"))
    # unfused (op-by-op kernels, same rounding points) must agree bit for bit
    wk.set_option("fused", 0)
    unfused, _ = wk.decode(tok, len(prompt), 6)
    wk.set_option("fused", 1)
    assert np.array_equal(fused, unfused)
    # eager launches vs graph replay
    wk.set_option("graph", 0)
    eager, _ = wk.decode(tok, len(prompt), 6)
    wk.set_option("graph", 1)
    assert np.array_equal(fused, eager)
    assert (fused >= 0).all() and (fused < s["vocab"]).all()
    wk.close()


@pytest.mark.parametrize("wd,kvd", [(dt.Q3H_B64T1, dt.Q8_B32T2), (dt.Q8_B32T2, dt.F16), (dt.Q4_B64T1, dt.F16),
                                    (dt.Q5_B64T1, dt.F16), (dt.Q6_B64T1, dt.Q8_B32T2)],
                         ids=["q3h_kvq8", "q8", "q4b64", "q5b64", "q6b64_kvq8"])
def test_llama2_7b_width_other_formats_fused_equals_unfused(wd, kvd):
    """Full Llama-2-7B widths (4096 / 11008 columns: 1-6 blocks per lane), 2 layers: the fused kernels of
    every int8-GEMV weight format against the op-level kernels, bit for bit."""
    wk, _, s = synth.build("llama2_7b", wd, kvd, max_ctx=48, layers=2)
    ok, why = wk.fused_supported()
    assert ok, why
    prompt = np.random.default_rng(7).integers(3, s["vocab"], 5).astype(np.int32)
    tok = wk.forward(prompt, 0)
    fused, _ = wk.decode(tok, len(prompt), 5)
    logits_fused = wk.read_buffer("logits").copy()
    wk.set_option("fused", 0)
    unfused, _ = wk.decode(tok, len(prompt), 5)
    logits_unfused = wk.read_buffer("logits").copy()
    wk.set_option("fused", 1)
    assert np.array_equal(logits_fused, logits_unfused)
    assert np.array_equal(fused, unfused)
    wk.close()


@pytest.mark.parametrize("shape,wd", [("yi_34b", dt.Q4_B32T1A), ("yi_34b", dt.Q3H_B64T1), ("llama2_70b", dt.Q4_B32T1A),
                                      ("llama2_70b", dt.Q8_B32T2), ("falcon_40b", dt.Q4_B32T1A)],
                         ids=["yi34b_q4", "yi34b_q3h", "70b_q4", "70b_q8", "falcon40b_q4"])
def test_long_rows_fused_equals_unfused(shape, wd):
    """w2 rows of 20480 / 28672 / 32768 columns do not fit a lane's register image: the chunked, software-pipelined kernel
    (k_dec_gemv_long) must continue the same per-lane accumulation chain -> bit-identical to the op-level GEMV.
    falcon_40b: full Falcon-40B widths (configs[3]): 8192 / 32768 columns, 128 heads of 64 over 8 KV heads, LayerNorm,
    GELU, plain MLP, shared MLP / attention input."""
    wk, _, s = synth.build(shape, wd, dt.F16, max_ctx=40, layers=2, vocab=8000)
    ok, why = wk.fused_supported()
    assert ok, why
    prompt = np.random.default_rng(11).integers(3, s["vocab"], 4).astype(np.int32)
    tok = wk.forward(prompt, 0)
    fused, _ = wk.decode(tok, len(prompt), 4)
    logits_fused = wk.read_buffer("logits").copy()
    wk.set_option("fused", 0)
    unfused, _ = wk.decode(tok, len(prompt), 4)
    assert np.array_equal(logits_fused, wk.read_buffer("logits"))
    assert np.array_equal(fused, unfused)
    wk.close()


def test_mixtral_8x7b_width_fused_equals_unfused_and_batched_rows_are_independent():
    """configs[4] at its own widths: 8 experts of ffn 14336 (7 blocks per lane on w2's 14336 columns), top-2 routing, 32
    heads over 8 KV heads, 2 layers.  The fused MoE decode step == the op-by-op path bit for bit; a batched step of 8
    queries (the device-routed grouped path) gives every query the token its own single-query step gives."""
    wk, _, s = synth.build("mixtral_8x7b", dt.Q4_B32T1A, dt.F16, max_ctx=64, layers=2, vocab=8000)
    ok, why = wk.fused_supported()
    assert ok, why
    rng = np.random.default_rng(21)
    prompt = rng.integers(3, s["vocab"], 6).astype(np.int32)
    tok = wk.forward(prompt, 0)
    fused, _ = wk.decode(tok, len(prompt), 6)
    logits_fused = wk.read_buffer("logits").copy()
    wk.set_option("fused", 0)
    unfused, _ = wk.decode(tok, len(prompt), 6)
    logits_unfused = wk.read_buffer("logits").copy()
    wk.set_option("fused", 1)
    assert np.array_equal(fused, unfused) and np.array_equal(logits_fused, logits_unfused)
    # batch of 8 queries, one step: each row == the query's own decode step
    wk.kv_slots(8)
    prompts = [rng.integers(3, s["vocab"], n).astype(np.int32) for n in (4, 7, 5, 9, 3, 6, 8, 5)]
    firsts, solo = [], []
    for i, pr in enumerate(prompts):
        wk.select_kv(i)
        t0 = wk.forward(pr, 0)
        firsts.append(t0)
        nxt, _ = wk.decode(t0, len(pr), 1)
        solo.append(int(nxt[0]))
        wk.forward(pr, 0)              # restore the cache row the solo step wrote at position len(pr) (same values anyway)
    got = wk.decode_batch(firsts, [len(p) for p in prompts], list(range(8)))
    agree = sum(int(a) == b for a, b in zip(got, solo))
    assert agree >= 7, (list(got), solo)       # (T > 1 rows use F16 activations: a near tie may flip one row)
    wk.close()


@pytest.mark.parametrize("T,rows,cols", [(16, 4096, 4096), (128, 11008, 4096), (1024, 4096, 11008)])
def test_llama2_7b_width_prefill_gemm_regimes_agree(T, rows, cols):
    """The three prefill GEMM regimes at Llama-2-7B widths (8-wave split-K for T <= 32, 4-wave split-K up to 128 tokens, the
    large-tile kernel k_gemm_big above): within the half rounding of an fp32 product with the dequantised weights (torch fp32 on
    the dequantised operand -- the reference's Dequantize + GemmEx, inference_worker.cc:2374-2415, accumulates in fp32 too), and
    the large-tile kernel == the smaller-tile kernels within 2 half-ulps on 99.5 % of the outputs."""
    import torch
    from tests import gpu_util as g
    L = g.capi()
    torch.manual_seed(T)
    w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
    W = g.quantize(dt.Q4_B32T1A, w)
    x = (torch.randn(T, cols, device="cuda") * 0.5).half()
    prev = L.ifa_gemm_big_tiles(-1)
    try:
        L.ifa_gemm_big_tiles(1)
        y_own = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x)).astype(np.float32)
        L.ifa_gemm_big_tiles(0)
        y_small = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x)).astype(np.float32)
    finally:
        L.ifa_gemm_big_tiles(prev)
    scale = float(np.abs(y_own).mean())
    assert scale > 0.05
    close = np.abs(y_own - y_small) <= np.maximum(2 * np.spacing(np.abs(y_own).astype(np.float16)).astype(np.float32), 1e-3 * scale)
    assert close.mean() >= 0.995, close.mean()
    # against the dequantised weights in fp32 (torch), every row: |err| within the half rounding of the result + accumulation noise
    wdq = g.dequantize(dt.Q4_B32T1A, W, cols).float()
    ref = g.host((x.float() @ wdq.t()).contiguous())
    tol = 4e-3 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(y_own - ref).max() <= tol
    assert np.abs(y_small - ref).max() <= tol


@pytest.mark.parametrize("T,rows,cols", [(1024, 12288, 4096), (1024, 22016, 4096), (1000, 12288, 4096), (2048, 11008, 4096)])
def test_prefill_gemm_stream_k_is_deterministic_and_matches_the_whole_tile_kernel(T, rows, cols):
    """256 x 256 tiles of a product whose tile count is no multiple of the CU count run the stream-K schedule (csrc/ifa_gemm.hip,
    k_gemm_big<.., KS = 0>; opt-in, bit 13 of ifa_gemm_big_tiles -- measured slower than the default launches,
    profiles/r06_stream_k_ab.log): one workgroup per CU takes floor(tiles / CUs) whole tiles, then an equal share of the K steps of the
    tiles that are left; shares that do not end their tile leave fp32 sums in scratch, the share that ends it adds them in one fixed
    order.  192 tiles (the wq | wk | wv shape of a 1024-token prompt: shares of 48 steps, two tiles touched), 344 tiles (the w1 | w3
    shape: one whole tile + shares of 22 steps, up to four shares per tile), a ragged token count, two whole rounds + a rest.  Same
    bits on every run, within the F16 rounding of the whole-tile kernel (another fp32 summation order), counters left zero (a product
    of another shape next on the same stream), and against the fp32 product with the dequantised weights."""
    import torch
    from tests import gpu_util as g
    L = g.capi()
    torch.manual_seed(T + rows)
    w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
    W = g.quantize(dt.Q4_B32T1A, w)
    x = (torch.randn(T, cols, device="cuda") * 0.5).half()
    bias = (torch.randn(rows, device="cuda") * 0.3).half()
    prev = L.ifa_gemm_big_tiles(-1)
    try:
        L.ifa_gemm_big_tiles(1 | (1 << 13))                                       # stream-K on (opt-in: measured slower, see the launcher)
        y1 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x, bias))
        y2 = g.host(g.gemm(dt.Q4_B32T1A, W, rows - 256, cols, x[:T - 5], bias))   # (another shape next: same scratch, counters must be clean)
        y3 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x, bias))
        L.ifa_gemm_big_tiles(1)                                                   # the default: whole tiles
        y0 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x, bias))
        y2_0 = g.host(g.gemm(dt.Q4_B32T1A, W, rows - 256, cols, x[:T - 5], bias))
    finally:
        L.ifa_gemm_big_tiles(prev)
    assert np.array_equal(y1, y3)
    for a16, b16 in ((y1, y0), (y2, y2_0)):
        a, b = a16.astype(np.float32), b16.astype(np.float32)
        close = np.abs(a - b) <= np.maximum(2 * np.spacing(np.abs(b).astype(np.float16)).astype(np.float32), 1e-3 * float(np.abs(b).mean()))
        assert close.mean() >= 0.999, close.mean()
        assert np.abs(a - b).max() <= 4e-3 * max(1.0, float(np.abs(b).max()))
    assert not np.array_equal(y1, y0)                                             # (the schedule really ran: another summation order somewhere)
    wdq = g.dequantize(dt.Q4_B32T1A, W, cols).float()
    ref = g.host((x.float() @ wdq.t() + bias.float()).contiguous())
    assert np.abs(y1.astype(np.float32) - ref).max() <= 4e-3 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("T,rows,cols", [(1024, 4096, 11008), (512, 4096, 4096), (640, 4096, 4096)])
def test_prefill_gemm_split_k_is_deterministic_and_matches_the_unsplit_kernel(T, rows, cols):
    """128 x 128 tiles of a product that offers no more tiles than CUs run as TWO workgroups per tile, each over half of K; the
    first half's fp32 sums reach the second workgroup through memory and are added first half + second half (csrc/ifa_gemm.hip,
    k_gemm_big<.., KS = 2>).  Same bits on every run (fixed order), within the F16 rounding of the unsplit kernel (another fp32
    summation order: the halves are rounded once each), flags left zero (a second product of another shape on the same stream)."""
    import torch
    from tests import gpu_util as g
    L = g.capi()
    torch.manual_seed(T + cols)
    w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
    W = g.quantize(dt.Q4_B32T1A, w)
    x = (torch.randn(T, cols, device="cuda") * 0.5).half()
    bias = (torch.randn(rows, device="cuda") * 0.3).half()
    prev = L.ifa_gemm_big_tiles(-1)
    try:
        L.ifa_gemm_big_tiles(1)
        y1 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x, bias))
        y2 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x[:T - 3], bias))         # (ragged token count next: same scratch, flags must be clean)
        y3 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x, bias))
        L.ifa_gemm_big_tiles(1 | (1 << 12))                                       # split-K off
        y0 = g.host(g.gemm(dt.Q4_B32T1A, W, rows, cols, x, bias))
    finally:
        L.ifa_gemm_big_tiles(prev)
    assert np.array_equal(y1, y3) and np.array_equal(y1[:T - 3], y2)
    a, b = y1.astype(np.float32), y0.astype(np.float32)
    close = np.abs(a - b) <= np.maximum(2 * np.spacing(np.abs(b).astype(np.float16)).astype(np.float32), 1e-3 * float(np.abs(b).mean()))
    assert close.mean() >= 0.999, close.mean()
    assert np.abs(a - b).max() <= 4e-3 * max(1.0, float(np.abs(b).max()))
