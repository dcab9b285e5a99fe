"""-m gpu: LAYER-WISE, teacher-forced parity at Llama-2-7B size (VERDICT r4 "Next" 1a) -- a test where errors cannot compound.

The whole-model comparisons (tests/test_gpu_fullsize_oracle.py) see the two sides only after 1, 4 or 32 layers have re-quantised
each other's half-ulp differences; a fault that only shows in layers 2..32 would hide inside the depth law.  Here every one of the
32 layers of both headline configurations (Q4_B32T1A + F16 KV, Q3H_B64T1 + Q8_B32T2 KV) runs ALONE on the device, through the
launches the bench times (graph replay of the fused QKV + attention launch, Wo, W1 / W3, W2), on the ORACLE's state:
  * oracle.Model decodes 8 teacher-forced steps and keeps every layer's input (orc_model_set_capture) and its K / V caches;
  * the oracle's K / V rows of ALL layers are copied into the worker's caches, the oracle's input of layer l at step i into the
    worker's layer-input buffer (options debug_hidden_in, debug_layer0 = l, debug_layers = 1), ONE decode step at position i runs;
  * the layer's output (buffer "x2") is compared with the oracle's input of layer l + 1 at that step, and the new K / V row the
    layer stored with the oracle's row i.
Bounds (both sides round the same F16 values; they differ by fp32 summation order -- wave64 orders against the restated CUDA lane
order -- so single values move by an ulp and, now and then, ONE int8 code of a re-quantised activation (Wo input, FFN input, W2
input: 127 levels per 32-value block) lands on the other side of a rounding tie).  Measured on MI355X, Q4 + F16 KV, 32 layers x 4
positions (profiles/r05_layerwise_parity.log): median max|d out| = 0.0003-0.0025 x std(layer output), 90 % of the pairs <= 0.0055,
8 of 128 pairs between 0.006 and 0.027 (single flips; largest in layers 0-4, whose output std is smallest).  The test holds
    every (layer, position):            max |d out| <= ONE_FLIP x std      F16 KV 0.04, Q8 KV 0.06   (a flip + the usual figure)
    every LAYER, its best position:     <= PER_LAYER x std                 F16 KV 0.006, Q8 KV 0.01  (a fault of one layer shows at
                                                                            every position; a flip at one or two of the four)
    all pairs, median:                  <= 0.004 x std / 0.01
so a layer-2..32-only discrepancy cannot hide behind the depth law of the whole-model tests.  Steps run from the last position down
so that the rows below the step's position are still the oracle's when the step runs."""
import numpy as np
import pytest

import oracle as o
from inferflow_amd import dtypes as dt, synth
from tests import gpu_util as g

pytestmark = pytest.mark.gpu
N_STEPS = 8
STEPS_CHECKED = (7, 4, 1, 0)


ROUTE_TIE = 0.004      # router gap p[2nd] - p[3rd] (F16 probabilities) below which an MoE layer may legitimately pick another expert


@pytest.mark.parametrize("wd,kvd,bound,per_layer,overall", [(dt.Q4_B32T1A, dt.F16, 0.04, 0.006, 0.004), (dt.Q3H_B64T1, dt.Q8_B32T2, 0.06, 0.01, 0.01)],
                         ids=["q4_kvf16", "q3h_kvq8"])
def test_every_layer_of_llama2_7b_alone_on_the_oracles_state(wd, kvd, bound, per_layer, overall):
    _layerwise("llama2_7b", wd, kvd, bound, per_layer, overall)


def test_every_layer_of_mixtral_8x7b_alone_on_the_oracles_state():
    """configs[4]'s model at FULL depth (32 mixture-of-experts layers, 8 experts, top-2, 32 heads over 8 KV heads): whole-model logits
    cannot be compared at this depth -- with random-init routers 55 of 56 rows meet a router near tie in one of their 32 layers
    and the two sides then run DIFFERENT experts (r05 run, profiles/r05_mixtral_routing_ties.log) -- but a layer alone has ONE
    routing decision per row: pairs whose oracle margin is a near tie are skipped (counted), every other layer must match."""
    try:
        import psutil
        avail = psutil.virtual_memory().available / 2 ** 30
    except Exception:      # noqa: BLE001
        avail = 1e9
    if avail < 56:
        pytest.skip("needs ~56 GB of host memory for the read-back model, %.0f available" % avail)
    _layerwise("mixtral_8x7b", dt.Q4_B32T1A, dt.F16, 0.04, 0.006, 0.004)


def _layerwise(shape_name, wd, kvd, bound, per_layer, overall):
    from tests.model_util import oracle_model_from_worker
    max_ctx = 64
    wk, _, s = synth.build(shape_name, wd, kvd, max_ctx=max_ctx)
    ok, why = wk.fused_supported()
    assert ok, why
    L, D = s["layers"], s["dim"]
    om = oracle_model_from_worker(wk, s, max_ctx, kvd)
    om.capture_layers(True)
    toks = np.random.default_rng(77).integers(3, s["vocab"], N_STEPS).astype(np.int32)
    io, margins = [], []
    for i in range(N_STEPS):
        om.forward(np.array([toks[i]], np.int32), i, want_logits=False)
        io.append(om.layer_io())                                     # [L + 1][D]
        margins.append(om.layer_margins())                           # [L] (2.0: dense layer)
    k_orc = [om.kv_rows(l, False, N_STEPS) for l in range(L)]
    v_orc = [om.kv_rows(l, True, N_STEPS) for l in range(L)]
    row_bytes = k_orc[0].shape[1]
    # the worker's caches = the oracle's, all layers (same row format: F16 rows, or Q8_B32T2 rows of 34-byte blocks)
    wk.reset()
    for l in range(L):
        wk.write_buffer("kcache", k_orc[l], layer=l)
        wk.write_buffer("vcache", v_orc[l], layer=l)
    wk.set_option("debug_hidden_in", 1)
    wk.set_option("debug_layers", 1)
    worst = (0.0, -1, -1)
    errs, ties = [], []
    kv_worst = 0.0
    code_flips = 0
    for i in STEPS_CHECKED:
        for l in range(L):
            if margins[i][l] < ROUTE_TIE:        # the top-k cut of this layer's router is a near tie: either choice is right
                ties.append((l, i, round(float(margins[i][l]), 5)))
                continue
            wk.set_option("debug_layer0", l)
            wk.write_buffer("x", io[i][l].view(np.uint16))
            wk.decode(int(toks[i]), i, 1)
            out = wk.read_buffer("x2").view(np.float16).astype(np.float32)
            ref = io[i][l + 1].astype(np.float32)
            err = float(np.abs(out - ref).max()) / float(ref.std())
            errs.append((err, l, i))
            if err > worst[0]:
                worst = (err, l, i)
            # the K / V row this step stored against the oracle's row i
            for name, orc_rows in (("kcache", k_orc[l]), ("vcache", v_orc[l])):
                p, n = wk.buffer(name, l)
                raw = wk.read_buffer(name, l, nbytes=(i + 1) * row_bytes)[i * row_bytes:(i + 1) * row_bytes]
                if kvd == dt.F16:
                    a = raw.view(np.float16).astype(np.float32); b = orc_rows[i].view(np.float16).astype(np.float32)
                    d = float(np.abs(a - b).max()) / float(b.std())
                    kv_worst = max(kv_worst, d)
                    assert d <= 0.004, ("F16 cache row", name, l, i, d)            # (an ulp or two of values up to ~4 std)
                else:
                    a = raw.reshape(-1, 34); b = orc_rows[i].reshape(-1, 34)
                    ca, cb = a[:, 2:].view(np.int8).astype(np.int32), b[:, 2:].view(np.int8).astype(np.int32)
                    assert np.abs(ca - cb).max() <= 1, ("Q8 cache codes", name, l, i)
                    code_flips += int((ca != cb).sum())
                    sa = a[:, :2].copy().view(np.float16).astype(np.float32); sb = b[:, :2].copy().view(np.float16).astype(np.float32)
                    assert np.abs(sa - sb).max() <= 0.002 * float(sb.max()), ("Q8 cache scales", name, l, i)
            # (the row is put back: the steps below this position read rows < i only, but a later parametrisation may not)
            wk.write_buffer("kcache", k_orc[l][i], layer=l, offset=i * row_bytes)
            wk.write_buffer("vcache", v_orc[l][i], layer=l, offset=i * row_bytes)
    by_step = {i: sorted(e for e, l, i2 in errs if i2 == i) for i in STEPS_CHECKED}
    print("layer-wise |d out| / std by position (median, 90 %%, max over the %d layers): %s" % (
        L, "; ".join("pos %d: %.5f %.5f %.5f" % (i, v[len(v) // 2], v[int(len(v) * 0.9)], v[-1]) for i, v in by_step.items())))
    print("layer-wise per-layer MINIMUM over the positions, worst three: %s; router near ties skipped (layer, position, margin): %s" % (
        sorted(((round(min([e for e, l2, i in errs if l2 == l] or [0.0]), 5), l) for l in range(L)), reverse=True)[:3], ties))
    print("layer-wise parity %s / %s: worst layer output %.5f x std (layer %d, position %d; bound %.4f); cache rows: %s"
          % (dt.name(wd), dt.name(kvd), worst[0], worst[1], worst[2], bound,
             ("worst F16 value %.5f x std" % kv_worst) if kvd == dt.F16 else ("%d int8 codes off by one of %d" % (code_flips, 2 * len(errs) * s["kv_heads"] * s["head_dim"]))))
    wk.set_option("debug_hidden_in", 0); wk.set_option("debug_layers", 0); wk.set_option("debug_layer0", 0)
    wk.close()
    bad = [(e, l, i) for e, l, i in errs if e > bound]
    assert not bad, "%d of %d (layer, position) pairs over the one-flip bound %.4f x std; worst: layer %d at position %d, %.5f" % (
        len(bad), len(errs), bound, worst[1], worst[2], worst[0])
    min_layer = {l: min([e for e, l2, i in errs if l2 == l] or [0.0]) for l in range(L)}
    off = {l: round(v, 5) for l, v in min_layer.items() if v > per_layer}
    assert not off, "layers whose BEST position still exceeds %.4f x std (a systematic difference of the layer, not a flip): %s" % (per_layer, off)
    assert float(np.median([e for e, l, i in errs])) <= overall
    assert len(ties) <= len(STEPS_CHECKED) * L // 8, "too many router near ties skipped: %s" % (ties,)
