"""Oracle vs the committed golden vectors (generated from the reference's own
src/common/quantization.h by tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle as o

QD = o.QUANT_DTYPES


def _src(golden, dt):
    key = "src_q4b16_f16" if dt == o.Q4_B16 else "src_f16"
    return golden[key].view(np.float16)


def test_f2h_matches_reference_half(golden):
    assert np.array_equal(o.f2h(golden["f2h_in"]).view(np.uint16), golden["f2h_out"])


def test_h2f_roundtrip_all_halfs():
    allh = np.arange(65536, dtype=np.uint16)
    f = o.h2f(allh)
    finite = np.isfinite(f)
    assert np.array_equal(o.f2h(f[finite]).view(np.uint16), allh[finite])
    assert np.array_equal(f[finite], allh.view(np.float16).astype(np.float32)[finite])


@pytest.mark.parametrize("dt", QD, ids=[o.DTYPE_NAMES[d] for d in QD])
def test_quantize_bit_exact(golden, dt):
    name = o.DTYPE_NAMES[dt]
    src = _src(golden, dt)
    # Q8_B32T2: the golden is the reference HOST routine; TensorOpr::Quantize on
    # the GPU path uses the device alg-2 kernel (appendix A3), tested below.
    got = o.quantize_q8_b32t2_host(src) if dt == o.Q8_B32T2 else o.quantize(dt, src)
    assert np.array_equal(got, golden["packed_" + name])
    if dt != o.Q8_B32T2:
        s32 = (src.astype(np.float32) * np.float32(1.0001)).astype(np.float32)
        assert np.array_equal(o.quantize(dt, s32), golden["packed32_" + name])


@pytest.mark.parametrize("dt", QD, ids=[o.DTYPE_NAMES[d] for d in QD])
def test_dequantize_bit_exact(golden, dt):
    name = o.DTYPE_NAMES[dt]
    cols = int(golden["cols"])
    packed = golden["packed_" + name]
    assert np.array_equal(o.dequantize(dt, packed, cols).view(np.uint16), golden["deq16_" + name])
    assert np.array_equal(o.dequantize(dt, packed, cols, out_f32=True).view(np.uint32),
                          golden["deq32_" + name].view(np.uint32))


@pytest.mark.parametrize("dt", o.GETINT4_DTYPES, ids=[o.DTYPE_NAMES[d] for d in o.GETINT4_DTYPES])
def test_getint4_words(golden, dt):
    name = o.DTYPE_NAMES[dt]
    cols = int(golden["cols"])
    words = o.codes_to_int4_words(o.unpack_codes(dt, golden["packed_" + name], cols))
    assert np.array_equal(words, golden["int4_" + name])


def test_block_sizes():
    # sizeof checks of the reference structs
    exp = {o.Q8_B32T1: 36, o.Q8_B32T2: 34, o.Q6_B64T1: 52, o.Q5_B64T1: 44, o.Q5_B32T1: 24,
           o.Q4_B16: 10, o.Q4_B32T1A: 20, o.Q4_B64T1: 36, o.Q3H_B64T1: 32, o.Q3_B32T1A: 16,
           o.Q2_B32T1A: 12}
    for dt, b in exp.items():
        assert o.block_bytes(dt) == b
    if o.ref_lib() is not None:
        for dt in QD:
            assert o.ref_lib().ref_block_bytes(dt) == o.block_bytes(dt)


def test_act_quant_device_vs_host_routine(golden):
    """Device alg-2 quantizer == host QuantizeRow_Q8_B32T2 wherever x/scale and
    x*(1/scale) round alike (the two differ only in that and in the zero
    threshold, appendix A3): scales identical, codes within 1."""
    src = golden["src_f16"].view(np.float16)
    dev = o.quantize_act_q8(src).reshape(src.shape[0], -1, 34)
    host = golden["packed_q8_b32t2"].reshape(src.shape[0], -1, 34)
    big = np.abs(src.astype(np.float32)).reshape(src.shape[0], -1, 32).max(-1) / 127 >= 1e-5
    assert np.array_equal(dev[big][:, :2], host[big][:, :2])
    d = dev[big][:, 2:].view(np.int8).astype(int) - host[big][:, 2:].view(np.int8).astype(int)
    assert np.abs(d).max() <= 1
    assert (d != 0).mean() < 0.01


def test_act_quant_semantics():
    rng = np.random.default_rng(3)
    x = rng.normal(0, 1, (3, 96)).astype(np.float16)
    x[1, :32] = 0
    x[2, 40] = 1000.0
    q = o.quantize_act_q8(x).reshape(3, 3, 34)
    scale = q[:, :, :2].copy().view(np.float16)[..., 0].astype(np.float32)
    codes = q[:, :, 2:].view(np.int8)
    xf = x.astype(np.float32).reshape(3, 3, 32)
    s32 = np.abs(xf).max(-1) / np.float32(127)
    assert np.array_equal(scale, s32.astype(np.float16).astype(np.float32))
    assert (codes[1, 0] == 0).all()
    with np.errstate(divide="ignore", invalid="ignore"):
        expect = np.where(s32[..., None] <= 1e-6, 0, np.round(xf / s32[..., None]))
    # np.round is half-to-even, roundf is half-away: exclude exact .5 ties
    tie = np.abs(np.abs(xf / np.maximum(s32[..., None], 1e-30)) % 1 - 0.5) < 1e-6
    assert np.array_equal(codes[~tie], np.clip(expect, -128, 127).astype(np.int8)[~tie])


def test_specialised_q4_gemv_loop_is_bit_identical_to_the_general_one():
    """orc_gemv_ax8 has a Q4_B32T1 loop without the fp64 shadow sum (what bench.py's cpu_baseline times); it must produce
    the same halves as the general per-block loop, which the golden fixtures and the GPU kernels are pinned to."""
    import os
    from oracle import oracle as om
    rng = np.random.default_rng(12)
    for d in (o.Q4_B32T1A, o.Q4_B32T1B):
        for rows, cols in ((7, 32), (33, 256), (64, 4096), (5, 11008)):
            w = rng.normal(0, 0.05, (rows, cols)).astype(np.float16)
            Wq = o.quantize(d, w)
            xq = o.quantize_act_q8(rng.normal(0, 1.5, (1, cols)).astype(np.float16)).reshape(-1)
            fast = np.asarray(o.gemv_ax8(d, Wq, rows, cols, xq)).view(np.uint16).copy()
            om.lib().orc_set_slow_paths(1)
            try:
                slow = np.asarray(o.gemv_ax8(d, Wq, rows, cols, xq)).view(np.uint16).copy()
            finally:
                om.lib().orc_set_slow_paths(0)
            assert np.array_equal(fast, slow), (d, rows, cols)
    assert 1 <= o.usable_cpus() <= (os.cpu_count() or 1)


def test_multi_row_product_is_the_per_row_product_bit_for_bit():
    """orc_gemm_f16x (the whole-model oracle's T > 1 steps: every weight row dequantised once) against orc_gemv_f16x row by row"""
    rng = np.random.default_rng(31)
    for d in (o.Q4_B32T1A, o.Q3H_B64T1, o.Q8_B32T2, o.Q6_B64T1, o.Q2_B32T1B, o.F16):
        rows, cols, T = 96, 256, 5
        w = (rng.standard_normal((rows, cols)) * 0.05).astype(np.float16)
        W = w.view(np.uint16) if d == o.F16 else o.quantize(d, w)
        X = rng.standard_normal((T, cols)).astype(np.float16)
        b = rng.standard_normal(rows).astype(np.float16)
        for bias in (None, b):
            got = o.gemm_f16x(d, W, rows, cols, X, bias)
            ref = np.stack([o.gemv_f16x(d, W, rows, cols, X[t], bias=bias) for t in range(T)])
            assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), (d, bias is None)
