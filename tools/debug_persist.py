#!/usr/bin/env python3
"""Persistent decode launch vs the five-launch layer, on one model: tokens and logits must be bit-identical.

    python tools/debug_persist.py <shape> [wdtype] [kv_dtype] [--steps N] [--bisect] [--trace LAYER] [--time N]

On a mismatch (or with --bisect) the step is re-run with the first N layers only (option debug_layers) for N = 1, 2, ...
and every hand-off of layer N - 1 (q|k|v, quantised attention output, attention + residual, gated product, layer
output) is compared with the buffers the five-launch path leaves.  --trace prints the per-phase stamps of one layer.
Each model runs in the calling process: run it under `timeout` (a broken hand-off gives up by itself after 20 ms)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from inferflow_amd import dtypes as dt, synth, worker as W


def build(shape, wdtype, kv_dtype, max_ctx):
    wk, _, s = synth.build(shape, wdtype, kv_dtype, max_ctx=max_ctx, quant_threshold=0)
    dev = "cuda:0"
    # non-trivial norm weights (synth uses ones)
    for layer in range(s["layers"]):
        for tid, seed in ((W.T_ATTN_NORM, 7000), (W.T_FFN_NORM, 8000)):
            w = (1.0 + synth.gen_f16((s["dim"],), seed + layer, 0.1, dev).float()).half()
            wk.set_tensor_f16(layer, tid, dt.F16, w, 1, s["dim"])
    wk.set_tensor_f16(-1, W.T_OUT_NORM, dt.F16, (1.0 + synth.gen_f16((s["dim"],), 6999, 0.1, dev).float()).half(), 1, s["dim"])
    return wk, s


def arena_views(wk, s):
    """granule arenas -> dict of (tags, values) per edge"""
    raw = wk.read_buffer("ps_arena").view(np.uint64)
    QD, KVD = s["heads"] * s["head_dim"], s["kv_heads"] * s["head_dim"]
    counts = [s["dim"] // 2, (QD + 2 * KVD) // 2, QD // 4 + QD // 16, s["dim"] // 2, s["ffn"] // 2]
    out, off = {}, 0
    for name, n in zip(("x", "qkv", "att", "a", "act"), counts):
        g = raw[off:off + n]
        out[name] = ((g >> np.uint64(32)).astype(np.uint32), (g & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        off += (n + 63) // 64 * 64
    return out


def halfs(v32):
    return v32.view(np.uint16)


def run_step(wk, tok, pos, persist, n_layers=0, steps=1):
    wk.set_option("persist", persist)
    wk.set_option("debug_layers", n_layers)
    toks, ms = wk.decode(tok, pos, steps)
    logits = wk.read_buffer("logits").view(np.uint16).copy()
    return toks, ms, logits


def bisect(wk, s, tok, pos):
    QD, KVD = s["heads"] * s["head_dim"], s["kv_heads"] * s["head_dim"]
    ok_all = True
    for n in range(1, s["layers"] + 1):
        run_step(wk, tok, pos, 0, n)
        ref = dict(dqkv=wk.read_buffer("dqkv").view(np.uint16).copy(), attq=wk.read_buffer("attq").copy(),
                   a=wk.read_buffer("a").view(np.uint16).copy(), t1=wk.read_buffer("t1").view(np.uint16).copy()[:s["ffn"]],
                   xo=wk.read_buffer("x2" if n % 2 == 1 else "x").view(np.uint16).copy())
        try:
            run_step(wk, tok, pos, 1, n)
        except Exception as e:  # noqa: BLE001
            print("  layers=%d: persistent launch failed: %s" % (n, e))
            ok_all = False
        ar = arena_views(wk, s)
        ep = lambda edge, layer=n - 1: layer * 8 + edge + 1  # noqa: E731
        rep = []
        # q | k | v
        tags, vals = ar["qkv"]
        rep.append(("qkv", int((tags != ep(1)).sum()), int((halfs(vals) != ref["dqkv"]).sum())))
        # quantised attention image: codes | scale | xsum
        tags, vals = ar["att"]
        codes_ref = ref["attq"][:QD].view(np.uint32)
        sc_off = (QD + 15) // 16 * 16
        scale_ref = ref["attq"][sc_off:sc_off + QD // 32 * 4].view(np.uint32)
        xsum_ref = ref["attq"][sc_off + QD // 32 * 4:sc_off + QD // 32 * 8].view(np.uint32)
        img_ref = np.concatenate([codes_ref, scale_ref, xsum_ref])
        rep.append(("att", int((tags != ep(2)).sum()), int((vals != img_ref).sum())))
        tags, vals = ar["a"]
        rep.append(("a", int((tags != ep(3)).sum()), int((halfs(vals) != ref["a"]).sum())))
        tags, vals = ar["act"]
        rep.append(("act", int((tags != ep(4)).sum()), int((halfs(vals) != ref["t1"]).sum())))
        xo = wk.read_buffer("x2").view(np.uint16)
        rep.append(("x_out", 0, int((xo != ref["xo"]).sum())))
        bad = [r for r in rep if r[1] or r[2]]
        print("  layers=%d:" % n, " ".join("%s[tags_bad=%d vals_bad=%d]" % r for r in rep), "OK" if not bad else "MISMATCH", flush=True)
        if bad:
            ok_all = False
            name = bad[0][0]
            if name in ("qkv", "a", "act"):
                tags, vals = ar[name]
                refv = {"qkv": ref["dqkv"], "a": ref["a"], "act": ref["t1"]}[name]
                got = halfs(vals)
                idx = np.nonzero(got != refv)[0][:12]
                print("    first bad %s elements:" % name, [(int(i), hex(int(got[i])), hex(int(refv[i]))) for i in idx])
                print("    bad tags at granules:", np.nonzero(tags != ep({"qkv": 1, "a": 3, "act": 4}[name]))[0][:16].tolist())
            break
    return ok_all


def print_trace(wk, s, tok, pos, layer):
    wk.set_option("persist_trace", layer)
    run_step(wk, tok, pos, 1, 0)
    tr = wk.read_buffer("ps_trace").view(np.int64).reshape(-1, 32)
    wk.set_option("persist_trace", -1)
    labels = ["start", "x_in", "quantA", "qkv_done", "attn_in", "attn_done", "att_in", "wo_done", "a_in", "quantD", "w13_done",
              "act_in", "quantE", "w2_done"]
    t0 = tr[:, 0][tr[:, 0] > 0].min()
    rel = (tr[:, :14] - t0) * 0.01
    rel[tr[:, :14] == 0] = np.nan
    print("trace of layer %d (us after the first workgroup's layer start; median / max over workgroups):" % layer)
    prev = 0.0
    for i, lab in enumerate(labels):
        col = rel[:, i]
        if np.all(np.isnan(col)):
            continue
        med, mx = float(np.nanmedian(col)), float(np.nanmax(col))
        print("  %-10s med %7.2f  max %7.2f   (+%.2f)" % (lab, med, mx, med - prev))
        if lab not in ("attn_in", "attn_done"):
            prev = med
    extra = ["partials", "bar1", "quantised", "b0_wait", "b0_data", "b0_done", "b1_wait", "b1_data", "b1_done", "b2_wait", "b2_data", "b2_done"]
    rel2 = (tr[:, 14:26] - t0) * 0.01
    rel2[tr[:, 14:26] == 0] = np.nan
    print("  op 0 detail:", " ".join("%s=%.2f" % (extra[i], float(np.nanmedian(rel2[:, i]))) for i in range(12) if not np.all(np.isnan(rel2[:, i]))))
    ld = (tr[:, 26:30] - t0) * 0.01
    print("  loader: qkv issue start %.2f, w13 issue start %.2f, w13 issued %.2f, w2 issued %.2f (medians); ring-full events per layer: median %d" % (
        float(np.median(ld[:, 0])), float(np.median(ld[:, 1])), float(np.median(ld[:, 2])), float(np.median(ld[:, 3])), int(np.median(tr[:, 25]))))
    ok = (tr[:, 30] > 0) & (tr[:, 31] > 0) & (tr[:, 13] > 0)
    if ok.any():
        mhz = (tr[ok, 31] - tr[ok, 30]) / ((tr[ok, 13] - tr[ok, 0]) * 0.01)
        print("  shader clock during the layer: median %.0f MHz (min %.0f, max %.0f)" % (float(np.median(mhz)), float(mhz.min()), float(mhz.max())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape")
    ap.add_argument("wdtype", nargs="?", default="q4")
    ap.add_argument("kv", nargs="?", default="f16")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--prompt", type=int, default=16)
    ap.add_argument("--bisect", action="store_true")
    ap.add_argument("--trace", type=int, default=-1)
    ap.add_argument("--time", type=int, default=0)
    ap.add_argument("--timeout-us", type=int, default=20000)
    ap.add_argument("--opt", action="append", default=[], help="name=value worker options (e.g. persist_depth=2)")
    a = ap.parse_args()
    wd = {"q4": dt.Q4_B32T1A, "q3h": dt.Q3H_B64T1}[a.wdtype]
    kv = {"f16": dt.F16, "q8": dt.Q8_B32T2}[a.kv]
    wk, s = build(a.shape, wd, kv, 1024)
    wk.set_option("persist_timeout_us", a.timeout_us)
    for o in a.opt:
        k, v = o.split("=")
        wk.set_option(k, int(v))
    prompt = (np.arange(a.prompt, dtype=np.int32) * 7 + 3) % s["vocab"]
    tok = int(wk.forward(prompt, 0))
    pos = a.prompt
    print("model %s %s kv %s: prompt %d tokens, first generated token %d" % (a.shape, a.wdtype, a.kv, a.prompt, tok), flush=True)
    t_ref, _, l_ref = run_step(wk, tok, pos, 0, 0, a.steps)
    ok = True
    try:
        t_ps, _, l_ps = run_step(wk, tok, pos, 1, 0, a.steps)
        same_t = bool(np.array_equal(t_ref, t_ps))
        same_l = bool(np.array_equal(l_ref, l_ps))
        print("tokens five-launch:", t_ref.tolist())
        print("tokens persistent :", t_ps.tolist())
        print("RESULT %s: tokens %s, last-step logits %s (%d of %d differ)" % (a.shape, "identical" if same_t else "DIFFER",
              "bit-identical" if same_l else "DIFFER", int((l_ref != l_ps).sum()), l_ref.size), flush=True)
        ok = same_t and same_l
    except Exception as e:  # noqa: BLE001
        print("RESULT %s: persistent decode FAILED: %s" % (a.shape, e), flush=True)
        ok = False
    if a.bisect or not ok:
        print("bisecting by layer count:", flush=True)
        ok = bisect(wk, s, tok, pos) and ok
    if a.trace >= 0:
        try:
            print_trace(wk, s, tok, pos, a.trace)
        except Exception as e:  # noqa: BLE001
            print("trace failed:", e)
    if a.time > 0:
        for persist in (0, 1, 0, 1):
            try:
                wk.set_option("persist", persist)
                wk.set_option("debug_layers", 0)
                wk.decode(tok, pos, 4)
                _, ms = wk.decode(tok, pos, a.time)
                print("TIME %s persist=%d: %.4f ms/token, %.1f tok/s" % (a.shape, persist, ms / a.time, 1000.0 * a.time / ms), flush=True)
            except Exception as e:  # noqa: BLE001
                print("TIME persist=%d failed: %s" % (persist, e))
        try:
            wk.set_option("persist", 1)
            us = wk.time_kernel(6, 20)
            print("persistent launch alone: %.1f us per token (%d layers) = %.2f us per layer" % (us, s["layers"], us / s["layers"]))
        except Exception as e:  # noqa: BLE001
            print("time_kernel(6) failed:", e)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
