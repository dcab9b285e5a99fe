"""-m gpu: op-level parity of the HIP kernels (through the C ABI) against the
CPU oracle.  Integer / byte results must be bit-exact; fp results within the
tolerance written next to each assert."""
import numpy as np
import pytest
import torch

import oracle as o
from inferflow_amd import dtypes as dt
import inferflow_amd as ia
from tests import gpu_util as g

pytestmark = pytest.mark.gpu

IDS = lambda ds: [dt.NAMES[d] for d in ds]  # noqa: E731


def _rows(rng, rows, cols, scale=0.05):
    x = rng.normal(0, scale, (rows, cols))
    x[0, : min(cols, 64)] = 0.0              # all-zero blocks
    if rows > 1:
        x[1, :] = 0.125                      # constant row
    if rows > 2:
        x[2, 5] = 30.0                       # outlier
    return x.astype(np.float16)


# ----------------------------------------------------------------- codecs
@pytest.mark.parametrize("d", dt.QUANT, ids=IDS(dt.QUANT))
def test_quantize_bit_exact_vs_oracle(d):
    rng = np.random.default_rng(d)
    src = _rows(rng, 33, 512)
    if d == dt.Q4_B16:
        src = np.clip(src.astype(np.float32), -0.9, 1.4).astype(np.float16)
    got = g.host(g.quantize(d, g.dev(src)))
    assert np.array_equal(got, o.quantize(d, src))
    s32 = (src.astype(np.float32) * np.float32(1.0001)).astype(np.float32)
    got32 = g.host(g.quantize(d, g.dev(s32)))
    assert np.array_equal(got32, o.quantize(d, s32))


@pytest.mark.parametrize("d", dt.QUANT, ids=IDS(dt.QUANT))
def test_quantize_and_dequantize_match_golden(golden, d):
    name = dt.NAMES[d]
    cols = int(golden["cols"])
    src = golden["src_q4b16_f16" if d == dt.Q4_B16 else "src_f16"].view(np.float16)
    if d != dt.Q8_B32T2:   # golden Q8_B32T2 is the host routine; device = alg 2 (appendix A3)
        assert np.array_equal(g.host(g.quantize(d, g.dev(src))), golden["packed_" + name])
    deq = g.host(g.dequantize(d, g.dev(golden["packed_" + name]), cols))
    assert np.array_equal(deq.view(np.uint16), golden["deq16_" + name])


def test_quantize_rejects_bad_shapes():
    x = torch.zeros((2, 48), dtype=torch.float16, device="cuda")
    out = g.empty_u8(2, 64)
    rc = g.capi().ifa_quantize(dt.Q4_B64T1, g.p(x), 2, 48, g.p(out), None)
    assert rc == -1 and b"multiple" in g.capi().ifa_last_error()
    assert g.capi().ifa_quantize(dt.F16, g.p(x), 2, 48, g.p(out), None) == -1
    assert g.capi().ifa_quantize(dt.Q4_B32T1A, g.p(x), 0, 64, g.p(out), None) == 0   # empty is fine


@pytest.mark.parametrize("cols", [32, 96, 100, 4096, 11008])
def test_act_quant_bit_exact(cols):
    rng = np.random.default_rng(cols)
    x = rng.normal(0, 1.0, (3, cols)).astype(np.float16)
    x[1, :32] = 0
    x[2, min(cols - 1, 40)] = 500.0
    got = g.host(g.quantize_act(g.dev(x)))
    exp = o.quantize_act_q8(x)
    if cols % 32:   # ragged tail: bytes past the valid lanes of the last block are unspecified
        nfull = cols // 32
        assert np.array_equal(got[:, : nfull * 34], exp[:, : nfull * 34])
        tail = slice(nfull * 34, nfull * 34 + 2 + cols % 32)
        assert np.array_equal(got[:, tail], exp[:, tail])
    else:
        assert np.array_equal(got, exp)


# ------------------------------------------------------------------- GEMV
def _check_gemv(y_gpu, y_orc, y64, tag):
    ulp = g.half_ulp_diff(y_gpu, y_orc)
    # both sides are fp32 sums of the same exact terms in different orders:
    # after rounding to half they may differ by at most 1 ulp, and rarely.
    assert ulp.max() <= 1, "%s: max ulp %d" % (tag, ulp.max())
    assert (ulp != 0).sum() <= max(1, 0.03 * ulp.size), "%s: %d of %d rows differ" % (tag, (ulp != 0).sum(), ulp.size)
    err = np.abs(y_gpu.astype(np.float64) - y64)
    tol = 2.0 ** -10 * np.abs(y64) + 2e-3 * np.abs(y64).mean() + 1e-6
    assert (err <= tol).all(), "%s: |gpu - f64| exceeds half rounding + fp32 slack" % tag


AX8_SHAPES = [(37, 256), (64, 4096), (9, 11008), (5, 64), (12, 1376)]


@pytest.mark.parametrize("d", dt.AX8, ids=IDS(dt.AX8))
@pytest.mark.parametrize("rows,cols", AX8_SHAPES)
def test_gemv_int8_path(d, rows, cols):
    if cols % dt.block_capacity(d):
        pytest.skip("cols not a multiple of the block capacity")
    rng = np.random.default_rng(rows * 131 + cols + d)
    w = rng.normal(0, 0.05, (rows, cols)).astype(np.float16)
    x = rng.normal(0, 1.0, (1, cols)).astype(np.float16)
    Wq = o.quantize(d, w)
    xq = o.quantize_act_q8(x)
    y_orc, y64 = o.gemv_ax8(d, Wq, rows, cols, xq, want_f64=True)
    Wd, xd = g.dev(Wq), g.dev(xq)
    y = g.host(g.gemv(d, Wd, rows, cols, xd, dt.Q8_B32T2))
    _check_gemv(y, y_orc, y64, "aos")
    yt = g.host(g.gemv_tiled(d, g.repack(d, Wd, rows, cols), rows, cols, xd))
    assert np.array_equal(yt.view(np.uint16), y.view(np.uint16)), "tiled layout must give identical bits"
    bias = rng.normal(0, 0.5, rows).astype(np.float16)
    yb = g.host(g.gemv(d, Wd, rows, cols, xd, dt.Q8_B32T2, g.dev(bias)))
    assert np.array_equal(yb.view(np.uint16), o.add(y, bias).view(np.uint16))


def test_gemv_int8_rejects_ineligible_weight_type():
    W = g.empty_u8(4, 128); x = g.empty_u8(1, 4 * 34); y = g.empty_f16(4)
    rc = g.capi().ifa_gemv(dt.Q3_B32T1A, g.p(W), 4, 128, dt.Q8_B32T2, g.p(x), None, g.p(y), None)
    assert rc == -1
    rc = g.capi().ifa_gemv(dt.Q4_B64T1, g.p(W), 4, 96, dt.Q8_B32T2, g.p(x), None, g.p(y), None)
    assert rc == -1    # cols % 64 (GemvCheckN, tensor_mul.cu:1123)


F16X = dt.QUANT + [dt.F16]


@pytest.mark.parametrize("d", F16X, ids=IDS(F16X))
@pytest.mark.parametrize("rows,cols", [(19, 256), (32, 4096), (7, 1000)])
def test_gemv_fp16_activation_path(d, rows, cols):
    cap = dt.block_capacity(d)
    if cols % cap:
        pytest.skip("cols not a multiple of the block capacity")
    rng = np.random.default_rng(rows + cols + d)
    w = rng.normal(0, 0.05, (rows, cols)).astype(np.float16)
    if d == dt.Q4_B16:
        w = np.clip(w.astype(np.float32), -0.9, 1.4).astype(np.float16)
    x = rng.normal(0, 1.0, cols).astype(np.float16)
    bias = rng.normal(0, 0.5, rows).astype(np.float16)
    Wq = w if d == dt.F16 else o.quantize(d, w)
    y_orc, y64 = o.gemv_f16x(d, Wq, rows, cols, x, bias=bias, want_f64=True)
    y = g.host(g.gemv(d, g.dev(Wq), rows, cols, g.dev(x), dt.F16, g.dev(bias)))
    # fp32 accumulation in a different order, then half rounding twice (sum, +bias)
    ulp = g.half_ulp_diff(y, y_orc)
    assert ulp.max() <= 2 and (ulp != 0).sum() <= max(2, 0.1 * ulp.size)
    assert np.allclose(y.astype(np.float64), y64, rtol=2e-3, atol=2e-3)


# ------------------------------------------------- full-size property checks
@pytest.mark.parametrize("rows,cols", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_gemv_q4_full_size_against_dequantised_fp32(rows, cols):
    """Llama-2-7B shapes (BASELINE configs[1]): the fused kernel must equal
    dequant(W) @ dequant(x) computed in fp32 by torch on the same device."""
    torch.manual_seed(rows + cols)
    w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
    x = torch.randn(1, cols, device="cuda").half()
    Wq = g.quantize(dt.Q4_B32T1A, w)
    xq = g.quantize_act(x)
    y = g.gemv(dt.Q4_B32T1A, Wq, rows, cols, xq, dt.Q8_B32T2).float()
    yt = g.gemv_tiled(dt.Q4_B32T1A, g.repack(dt.Q4_B32T1A, Wq, rows, cols), rows, cols, xq).float()
    assert torch.equal(y, yt)
    Wd = g.dequantize(dt.Q4_B32T1A, Wq, cols)                       # half-rounded values
    xs = xq.view(-1, 34)
    xscale = xs[:, :2].contiguous().view(torch.float16).float()
    xd = (xs[:, 2:].contiguous().view(torch.int8).float() * xscale).reshape(-1)
    # exact value uses unrounded dequantised weights; Wd is rounded to half, so allow that
    ref = (Wd.double() @ xd.double()).float()
    err = (y - ref).abs()
    bound = 2.0 ** -10 * ref.abs() + 4e-3 * ref.abs().mean()
    assert bool((err <= bound).all()), float((err / bound).max())
    # quantize -> dequantize round trip stays within half a step (+ fp16 rounding of base/scale)
    blocks = w.float().view(rows, -1, 32)
    step = (blocks.max(-1).values - blocks.min(-1).values) / 15
    rt = (Wd.float().view(rows, -1, 32) - blocks).abs().max(-1).values
    assert bool((rt <= 0.5 * step * 1.02 + 2e-4).all())


# --------------------------------------------------------- norms / elementwise
def _half_ulp_distance(a, b):
    """distance in units of the last place between two float16 arrays (monotonic integer view)"""
    def key(v):
        u = v.view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u)
    return np.abs(key(a) - key(b))


@pytest.mark.parametrize("cols", [288, 4096, 1000, 8192 + 24])
def test_rmsnorm_within_one_half_ulp_of_reference_order(cols):
    """The reference sums x*x over 128 strided per-thread chunks and a serial chain (unary_tensor_opr.h:216-289, restated
    by the oracle); the HIP kernels -- op level and fused prologues alike -- use one butterfly order (ifa_math.h).  Same
    fp32 quantity, different addition order: the scale may differ in its last fp32 bits, so a normalised half value may
    land on the neighbouring half.  Stated bound: every element within ONE half ulp of the oracle, >= 99.5 % identical,
    and every element within 0.501 half ulp of the fp64 value."""
    rng = np.random.default_rng(cols)
    x = rng.normal(0, 1.0, (3, cols)).astype(np.float16)
    w = rng.normal(1, 0.1, cols).astype(np.float16)
    b = rng.normal(0, 0.1, cols).astype(np.float16)
    y = g.empty_f16(3, cols)
    for (wv, bv, mb) in [(w, None, 0.0), (w, b, 0.0), (None, None, 0.0), (w, None, 1.0)]:
        ia.check(g.capi().ifa_layernorm(0, g.p(g.dev(x)), 3, cols, g.p(g.dev(wv)) if wv is not None else None,
                                        g.p(g.dev(bv)) if bv is not None else None, mb, 1e-5, g.p(y), g.stream()))
        exp = o.rmsnorm(x, wv, bv, multi_base=mb)
        got = g.host(y)
        d = _half_ulp_distance(got, exp)
        assert d.max() <= 1 and (d == 0).mean() >= 0.995, (int(d.max()), float((d == 0).mean()))
        x64 = x.astype(np.float64)
        t = x64 / np.sqrt((x64 * x64).mean(1, keepdims=True) + 1e-5)
        mag = np.abs(t)
        if wv is not None:
            t = t * (mb + wv.astype(np.float64))
            mag = np.abs(t)
            if bv is not None:
                t = t + bv.astype(np.float64)
                mag = mag + np.abs(bv.astype(np.float64))          # a + b may cancel: the fp32 roundings scale with |a| + |b|
        ulp = np.spacing(np.abs(got).astype(np.float16)).astype(np.float64)
        # the final half rounding (0.5 ulp) + the fp32 roundings of the intermediate steps (a few 2^-24 of the operands)
        assert (np.abs(got.astype(np.float64) - t) <= 0.501 * np.maximum(ulp, 2.0 ** -24) + 4 * 2.0 ** -24 * mag).all()


@pytest.mark.parametrize("cols", [8192, 200])
def test_stdnorm_bit_exact(cols):
    rng = np.random.default_rng(cols)
    x = rng.normal(0.3, 1.0, (2, cols)).astype(np.float16)
    w = rng.normal(1, 0.1, cols).astype(np.float16)
    b = rng.normal(0, 0.1, cols).astype(np.float16)
    y = g.empty_f16(2, cols)
    ia.check(g.capi().ifa_layernorm(1, g.p(g.dev(x)), 2, cols, g.p(g.dev(w)), g.p(g.dev(b)), 0.0, 1e-5, g.p(y), g.stream()))
    assert np.array_equal(g.host(y).view(np.uint16), o.stdnorm(x, w, b).view(np.uint16))


@pytest.mark.parametrize("order,partial", [(2, 1.0), (1, 1.0), (2, 0.5)])
def test_rope(order, partial):
    rng = np.random.default_rng(order)
    x = rng.normal(0, 1.0, (3, 4, 128)).astype(np.float16)
    xd = g.dev(x)
    ia.check(g.capi().ifa_rope(g.p(xd), 128, 4, 3, 57, 10000.0, order, partial, g.stream()))
    exp = o.rope(x, 57, 10000.0, order, partial)
    # device powf/cosf/sinf vs libm: angles up to ~60 rad, |x| ~ 1 -> abs error well under 2 half ulps at 1.0
    assert np.abs(g.host(xd).astype(np.float32) - exp.astype(np.float32)).max() <= 4e-3
    if partial < 1.0:   # untouched tail
        assert np.array_equal(g.host(xd)[..., 64:].view(np.uint16), x[..., 64:].view(np.uint16))


def test_softmax_masked():
    rng = np.random.default_rng(5)
    s = rng.normal(0, 2.0, (2, 3, 77)).astype(np.float16)
    sd = g.dev(s)
    ia.check(g.capi().ifa_softmax(g.p(sd), 77, 3, 2, 40, 1.5, g.stream()))
    exp = o.softmax(s, 40, 1.5)
    got = g.host(sd)
    assert np.abs(got.astype(np.float32) - exp.astype(np.float32)).max() <= 2e-3   # device expf vs libm
    assert (got[:, 0, 41:] == 0).all() and (got[:, 2, 43:] == 0).all()
    assert np.allclose(got.astype(np.float32).sum(-1), 1.0, atol=5e-3)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_activation(kind):
    rng = np.random.default_rng(kind)
    x = rng.normal(0, 2.0, (4, 256)).astype(np.float16)
    y = g.empty_f16(4, 256)
    ia.check(g.capi().ifa_activation(kind, 0, g.p(g.dev(x)), 4, 256, g.p(y), g.stream()))
    exp = o.act(x, kind)
    assert g.half_ulp_diff(g.host(y), exp).max() <= 1      # device expf/tanhf vs libm
    y2 = g.empty_f16(4, 128)
    ia.check(g.capi().ifa_activation(0, 1, g.p(g.dev(x)), 4, 128, g.p(y2), g.stream()))
    assert g.half_ulp_diff(g.host(y2), o.act(x, 0, is_glu=True)).max() <= 1


def test_add_mul_scale_bit_exact():
    rng = np.random.default_rng(9)
    a = rng.normal(0, 1.0, (4, 512)).astype(np.float16)
    b = rng.normal(0, 1.0, (4, 512)).astype(np.float16)
    c = g.empty_f16(4, 512)
    ia.check(g.capi().ifa_add(g.p(g.dev(a)), g.p(g.dev(b)), a.size, 0, g.p(c), g.stream()))
    assert np.array_equal(g.host(c).view(np.uint16), o.add(a, b).view(np.uint16))
    ia.check(g.capi().ifa_add(g.p(g.dev(a)), g.p(g.dev(b[0])), a.size, 512, g.p(c), g.stream()))
    assert np.array_equal(g.host(c).view(np.uint16), o.add(a, b[0], 512).view(np.uint16))
    ia.check(g.capi().ifa_mul(g.p(g.dev(a)), g.p(g.dev(b)), a.size, g.p(c), g.stream()))
    assert np.array_equal(g.host(c).view(np.uint16), o.mul(a, b).view(np.uint16))
    ia.check(g.capi().ifa_scale(g.p(g.dev(a)), 0.37, a.size, g.p(c), g.stream()))
    assert np.array_equal(g.host(c).view(np.uint16), o.scale(a, 0.37).view(np.uint16))


def test_argmax_first_max_wins():
    v = np.zeros(32000, np.float16)
    v[[77, 31999, 500]] = [3.0, 3.0, 2.0]
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    ia.check(g.capi().ifa_argmax(g.p(g.dev(v)), v.size, g.p(out), g.stream()))
    assert int(out.item()) == 77


# ---------------------------------------------------------------- attention
@pytest.mark.parametrize("kv_dtype", [dt.F16, dt.Q8_B32T2], ids=["kv_f16", "kv_q8"])
@pytest.mark.parametrize("heads,kv_heads,hd,n_ctx,qt,alibi", [
    (8, 8, 64, 37, 1, 0), (8, 2, 128, 200, 1, 0), (4, 4, 32, 16, 16, 0), (8, 4, 64, 50, 3, 1),
    # prefill tiles: the [32 x keys] score tile in LDS (300 keys) and, past ~2300 keys, in the global workspace (2600 keys,
    # a chunk of 40 queries behind a 2560-token prefix)
    (4, 2, 64, 300, 70, 0), (2, 1, 128, 2600, 40, 0), (2, 2, 64, 2500, 33, 1),
    # chunks of >= 64 queries: the two-pass kernels (64 queries per workgroup with the key blocks split over wave pairs; the
    # 128-query variant and the staged kernel on the same inputs below) -- whole prompt, a ragged last tile behind a prefix,
    # ALiBi, a chunk behind 2400 cached tokens
    (4, 2, 64, 300, 300, 0), (2, 1, 128, 420, 260, 0), (2, 2, 64, 700, 130, 1), (2, 1, 128, 2560, 160, 0)])
def test_attention(kv_dtype, heads, kv_heads, hd, n_ctx, qt, alibi):
    rng = np.random.default_rng(heads * 7 + n_ctx)
    prefix = n_ctx - qt
    q = rng.normal(0, 1.0, (qt, heads, hd)).astype(np.float16)
    k = rng.normal(0, 1.0, (n_ctx, kv_heads * hd)).astype(np.float16)
    v = rng.normal(0, 1.0, (n_ctx, kv_heads * hd)).astype(np.float16)
    if kv_dtype == dt.Q8_B32T2:
        kc, vc = o.quantize_act_q8(k), o.quantize_act_q8(v)
    else:
        kc, vc = k, v
    kq_scale = 1.0 if alibi else 2.0
    exp = o.attention(q, kc, vc, kv_dtype, n_ctx, prefix, heads, kv_heads, hd, kq_scale, bool(alibi), 0, heads)
    out = g.empty_f16(qt, heads * hd)
    ia.check(g.capi().ifa_attention(g.p(g.dev(q)), g.p(g.dev(kc)), g.p(g.dev(vc)), kv_dtype, n_ctx, qt, prefix,
                                    heads, kv_heads, hd, kq_scale, alibi, 0, heads, g.p(out), g.stream()))
    # P differs by expf implementation and sum order (<= 1 half ulp per weight); O is a convex mix of |v| ~ 1
    assert np.abs(g.host(out).astype(np.float32) - exp.astype(np.float32)).max() <= 6e-3
    if qt >= 64:        # the other kernels on the same input: same rounding points, other sum orders
        L = g.capi()
        variants = [("staged", lambda: L.ifa_attention_two_pass_min(0), lambda p: L.ifa_attention_two_pass_min(p))]
        if qt >= 128:
            variants.append(("128-query", lambda: L.ifa_attention_two_pass_min_keys(0), lambda p: L.ifa_attention_two_pass_min_keys(p)))
        for name, switch, restore in variants:
            prev = switch()
            try:
                out2 = g.empty_f16(qt, heads * hd)
                ia.check(L.ifa_attention(g.p(g.dev(q)), g.p(g.dev(kc)), g.p(g.dev(vc)), kv_dtype, n_ctx, qt, prefix,
                                         heads, kv_heads, hd, kq_scale, alibi, 0, heads, g.p(out2), g.stream()))
            finally:
                restore(prev)
            assert np.abs(g.host(out2).astype(np.float32) - exp.astype(np.float32)).max() <= 6e-3, name
            assert np.abs(g.host(out2).astype(np.float32) - g.host(out).astype(np.float32)).max() <= 4e-3, name


# ------------------------------------------------------------ prefill GEMM (MFMA)
GEMM_TYPES = [dt.Q4_B32T1A, dt.Q3H_B64T1, dt.Q8_B32T2, dt.Q6_B64T1, dt.Q2_B32T1B, dt.Q4_B16, dt.F16]


@pytest.mark.parametrize("d", GEMM_TYPES, ids=IDS(GEMM_TYPES))
@pytest.mark.parametrize("T,rows,cols", [(5, 70, 256), (64, 128, 1024), (33, 200, 4096), (130, 96, 192)])
def test_gemm_prefill_matches_per_token_oracle(d, T, rows, cols):
    """Y[t] must equal the reference's dequantise-to-half + fp32-accumulate product per token
    (orc_gemv_f16x); the MFMA sums the same exact products in a different order."""
    if cols % dt.block_capacity(d):
        pytest.skip("cols not a multiple of the block capacity")
    rng = np.random.default_rng(T + rows + cols + d)
    w = rng.normal(0, 0.05, (rows, cols)).astype(np.float16)
    if d == dt.Q4_B16:
        w = np.clip(w.astype(np.float32), -0.9, 1.4).astype(np.float16)
    x = rng.normal(0, 1.0, (T, cols)).astype(np.float16)
    bias = rng.normal(0, 0.5, rows).astype(np.float16)
    Wq = w if d == dt.F16 else o.quantize(d, w)
    y = g.host(g.gemm(d, g.dev(Wq), rows, cols, g.dev(x), g.dev(bias)))
    for t in (0, T // 2, T - 1):
        y_orc, y64 = o.gemv_f16x(d, Wq, rows, cols, x[t], bias=bias, want_f64=True)
        # different fp32 summation order: <= 2 half ulps, except where cancellation leaves |y| tiny
        ulp = g.half_ulp_diff(y[t], y_orc)
        small = np.abs(y[t].astype(np.float32) - y_orc.astype(np.float32)) <= 1e-3 * float(np.abs(y64).mean() + 1e-6)
        assert ((ulp <= 2) | small).all(), (t, ulp.max())
        assert ((ulp != 0) & ~small).sum() <= max(2, 0.1 * ulp.size)
        assert np.allclose(y[t].astype(np.float64), y64, rtol=2e-3, atol=2e-3 * float(np.abs(y64).mean() + 1))
    # no bias, and T == 1 degenerate tile
    y1 = g.host(g.gemm(d, g.dev(Wq), rows, cols, g.dev(x[:1])))
    y_orc = o.gemv_f16x(d, Wq, rows, cols, x[0])
    assert g.half_ulp_diff(y1[0], y_orc).max() <= 1


BIG_DT = [dt.Q4_B32T1A, dt.Q4_B32T1B, dt.Q8_B32T2, dt.Q5_B32T1, dt.Q4_B16, dt.F16]


@pytest.mark.parametrize("T,rows,cols", [(150, 320, 1024), (300, 136, 512), (257, 520, 192), (513, 264, 2112)])
@pytest.mark.parametrize("d", BIG_DT, ids=IDS(BIG_DT))
def test_gemm_large_tile_kernel_matches_per_token_oracle(d, T, rows, cols):
    """Above 128 tokens ifa_gemm runs the large-tile kernel (csrc/ifa_gemm.hip, k_gemm_big: the weight tile dequantised once
    per workgroup and K step into swizzled LDS, the activation tile by direct-to-LDS loads, 256 x 256 / 128 x 256 /
    128 x 128 output tiles, output rows written from an LDS tile).  Every tile shape on the same problem -- ragged token
    and row counts, a K of 3 and of 33 steps, with and without bias -- against the oracle's per-token product (MatrixMultiplication's
    T > 1 branch, src/transformer/inference_worker.cc:2374-2415), and bit-identical to each other."""
    L = g.capi()
    rng = np.random.default_rng(5 + d + T)
    w = rng.normal(0, 0.05, (rows, cols)).astype(np.float16)
    x = rng.normal(0, 1.0, (T, cols)).astype(np.float16)
    bias = rng.normal(0, 0.5, rows).astype(np.float16)
    Wq = w if d == dt.F16 else o.quantize(d, w)
    Wd, xd, bd = g.dev(Wq), g.dev(x), g.dev(bias)
    prev_big = L.ifa_gemm_big_tiles(-1)
    ys = {}
    try:
        for name, mode in (("auto", 1), ("256x256", 1 | (1 << 8)), ("128x256", 1 | (2 << 8)), ("128x128", 1 | (3 << 8)), ("small", 0)):
            L.ifa_gemm_big_tiles(mode)
            ys[name] = g.host(g.gemm(d, Wd, rows, cols, xd, bd))
        L.ifa_gemm_big_tiles(1)
        y_nobias = g.host(g.gemm(d, Wd, rows, cols, xd))
    finally:
        L.ifa_gemm_big_tiles(prev_big)
    for t in (0, 127, 128, T // 2, T - 1):
        y_orc, y64 = o.gemv_f16x(d, Wq, rows, cols, x[t], bias=bias, want_f64=True)
        ulp = g.half_ulp_diff(ys["auto"][t], y_orc)
        small = np.abs(ys["auto"][t].astype(np.float32) - y_orc.astype(np.float32)) <= 1e-3 * float(np.abs(y64).mean() + 1e-6)
        assert ((ulp <= 2) | small).all(), (t, ulp.max())
        assert (g.half_ulp_diff(y_nobias[t], o.gemv_f16x(d, Wq, rows, cols, x[t])) <= 2).mean() >= 0.98
    for name in ("256x256", "128x256", "128x128"):
        assert np.array_equal(ys[name], ys["auto"]), name          # same products in the same order whatever the tile
    assert (g.half_ulp_diff(ys["small"], ys["auto"]) <= 2).mean() >= 0.995


@pytest.mark.parametrize("T,rows,cols", [(2, 70, 256), (3, 128, 4096), (4, 200, 1024), (7, 96, 2048), (8, 64, 11008), (12, 100, 4096), (16, 48, 5120),
                                          (5, 4500, 384), (8, 64, 2080)])
@pytest.mark.parametrize("d", [dt.Q4_B32T1A, dt.Q4_B32T1B], ids=IDS([dt.Q4_B32T1A, dt.Q4_B32T1B]))
def test_gemm_rows_streaming_kernel_matches_per_token_oracle(d, T, rows, cols):
    """The 2..16-row weight-streaming GEMMs of the dynamic-batching step (tiled layout): csrc/ifa_gemm_rows_mfma.hip
    (matrix cores, row length % 128 == 0, several tiles per workgroup at 4500 rows, a partial LDS chunk at 5120 columns) and
    csrc/ifa_gemm_rows.hip (2..8 rows, any row length % 32: the 2080-column case) -- same contract as ifa_gemm: weights
    dequantised to half, half activations, fp32 accumulation, one F16 rounding."""
    import ctypes as C
    L = g.capi()
    L.ifa_gemm_rows_q4.restype = C.c_int
    L.ifa_gemm_rows_q4.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(T + rows + cols + d)
    w = rng.normal(0, 0.05, (rows, cols)).astype(np.float16)
    x = rng.normal(0, 1.0, (T, cols)).astype(np.float16)
    bias = rng.normal(0, 0.5, rows).astype(np.float16)
    Wq = o.quantize(d, w)
    Wt = g.repack(d, g.dev(Wq), rows, cols)
    for b in (bias, None):
        y = g.empty_f16(T, rows)
        rc = L.ifa_gemm_rows_q4(g.p(Wt), rows, cols, g.p(g.dev(x)), T, g.p(g.dev(b)) if b is not None else None, g.p(y), g.stream())
        assert rc == 0
        y = g.host(y)
        for t in range(T):
            y_orc, y64 = o.gemv_f16x(d, Wq, rows, cols, x[t], bias=b, want_f64=True)
            ulp = g.half_ulp_diff(y[t], y_orc)
            small = np.abs(y[t].astype(np.float32) - y_orc.astype(np.float32)) <= 1e-3 * float(np.abs(y64).mean() + 1e-6)
            assert ((ulp <= 2) | small).all(), (t, ulp.max())
            assert np.allclose(y[t].astype(np.float64), y64, rtol=2e-3, atol=2e-3 * float(np.abs(y64).mean() + 1))


def test_add_by_row_index_is_a_half_fma_scatter():
    # AddByRowIdx_Kernel (binary_tensor_opr.h:80-125): B[idx[r]] = hfma(A[r], w[r], B[idx[r]])
    rng = np.random.default_rng(9)
    A = rng.normal(0, 1, (5, 300)).astype(np.float16)
    B = rng.normal(0, 1, (7, 300)).astype(np.float16)
    idx = np.array([6, 0, 3, 2, 5], np.int32)
    w = rng.uniform(0, 1, 5).astype(np.float16)
    Bd = g.dev(B)
    ia.check(g.capi().ifa_add_by_row_index(g.p(Bd), g.p(g.dev(A)), 5, 300, g.p(g.dev(idx)), g.p(g.dev(w)), g.stream()))
    expect = B.copy()
    for r in range(5):   # exact fma in double, one rounding to half
        expect[idx[r]] = (A[r].astype(np.float64) * float(w[r]) + B[idx[r]].astype(np.float64)).astype(np.float16)
    assert np.array_equal(g.host(Bd).view(np.uint16), expect.view(np.uint16))


# ------------------------------------------------- MoE routing / KV store exports
def test_moe_route_topk_matches_oracle_rows():
    """ifa_moe_route_topk: the per-row part of BuildRowsForMoE on the device (host_tensor_opr.cc:190-244) -- experts and
    weights bit for bit with the oracle's restatement, incl. the < 1e-5 drop rule and the renormalisation switch."""
    rng = np.random.default_rng(8)
    for E, k, norm in [(8, 2, 1), (4, 2, 0), (16, 4, 1), (3, 3, 1)]:
        T = 37
        logits = rng.normal(0, 3.0, (T, E)).astype(np.float32)
        logits[5] = np.array([30.0] + [-30.0] * (E - 1), np.float32)          # one-hot row: the other experts fall below 1e-5
        probs = (np.exp(logits - logits.max(1, keepdims=True)) / np.exp(logits - logits.max(1, keepdims=True)).sum(1, keepdims=True)).astype(np.float16)
        sel = torch.zeros((T, k), dtype=torch.int32, device="cuda")
        wts = torch.zeros((T, k), dtype=torch.float16, device="cuda")
        ia.check(g.capi().ifa_moe_route_topk(g.p(g.dev(probs)), T, E, k, norm, g.p(sel), g.p(wts), g.stream()))
        sel_h, w_h = sel.cpu().numpy(), g.host(wts)
        for t in range(T):
            idx, w = o.moe_topk(probs[t].astype(np.float32), k, bool(norm))
            order = np.argsort(idx, kind="stable")
            exp_sel = [int(idx[i]) for i in order] + [-1] * (k - len(idx))
            exp_w = [np.float16(w[i]) for i in order] + [np.float16(0)] * (k - len(idx))
            assert sel_h[t].tolist() == exp_sel, (E, k, t)
            assert np.array_equal(w_h[t].view(np.uint16), np.array(exp_w, np.float16).view(np.uint16)), (E, k, t)
        assert sel_h[5].tolist()[0] == 0 and (k == 1 or sel_h[5][1] == -1)


@pytest.mark.parametrize("kvd", [dt.F16, dt.Q8_B32T2], ids=["f16", "q8"])
def test_kv_store_rows(kvd):
    """ifa_kv_store = LayerKVCache::SetKRows / SetVRows (kv_cache.cc:159-249): F16 rows copied, Q8 rows through the Alg2
    quantiser -- the cache bytes the oracle's quantiser produces, at the requested row offset, neighbours untouched."""
    rng = np.random.default_rng(3)
    T, kv_dim, max_ctx, row0 = 5, 256, 16, 7
    rows = rng.normal(0, 1.0, (T, kv_dim)).astype(np.float16)
    rb = dt.row_bytes(kvd, kv_dim)
    cache = torch.full((max_ctx * rb,), 0xAB, dtype=torch.uint8, device="cuda")
    ia.check(g.capi().ifa_kv_store(kvd, g.p(g.dev(rows)), T, kv_dim, g.p(cache), row0, g.stream()))
    got = cache.cpu().numpy().reshape(max_ctx, rb)
    exp = rows.view(np.uint8).reshape(T, rb) if kvd == dt.F16 else o.quantize_act_q8(rows).reshape(T, rb)
    assert np.array_equal(got[row0:row0 + T], exp)
    assert (got[:row0] == 0xAB).all() and (got[row0 + T:] == 0xAB).all()
