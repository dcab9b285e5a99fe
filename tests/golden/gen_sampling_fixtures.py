#!/usr/bin/env python3
"""Golden draws of the reference's OWN decoding strategies (oracle/_ref/ifa_ref_sampling = the reference's
sampling_strategy.cc / decoding_strategies.cc / sslib compiled from /root/reference by `make -C oracle ref_sampling`, under
our driver oracle/ref_sampling_driver.cc) -> tests/golden/ref_sampling.npz.  Runs in the build container only (needs
/root/reference); the fixture is data: logits rows, options, and the (id, weight) the reference selected draw after draw.

    python tests/golden/gen_sampling_fixtures.py
"""
import json, os, struct, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BIN = os.path.join(ROOT, "oracle", "_ref", "ifa_ref_sampling")
# SamplingStrategyId (src/transformer/sampling_strategy.h:54-67)
STD, GREEDY, TOP_K, TOP_P, FSD, RANDOM_FSD, MIN_P, TFS, TYPICAL, MIROSTAT = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10


def run_ref(logits, strategy, seed, temperature, n_draws, text=(), config=""):
    lg = np.ascontiguousarray(logits, np.float16)
    cfg = config.encode()
    with tempfile.TemporaryDirectory() as d:
        fi, fo = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fi, "wb") as f:
            f.write(struct.pack("<iiiifiii", 0x49465331, lg.size, strategy, seed, temperature, n_draws, len(text), len(cfg)))
            f.write(lg.view(np.uint16).tobytes())
            f.write(np.asarray(text, np.int32).tobytes())
            f.write(cfg)
        subprocess.run([BIN, fi, fo], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        raw = open(fo, "rb").read()
    n = struct.unpack_from("<i", raw, 0)[0]
    rec = np.frombuffer(raw, dtype=[("id", "<i4"), ("w", "<f4")], count=n, offset=4)
    pn = struct.unpack_from("<i", raw, 4 + 8 * n)[0]
    pool = np.frombuffer(raw, dtype=[("id", "<i4"), ("w", "<f4")], count=pn, offset=8 + 8 * n)
    return rec["id"].copy(), rec["w"].copy(), pool["id"].copy(), pool["w"].copy()


def cases():
    rng = np.random.default_rng(20240927)
    out = []
    for strategy in (STD, GREEDY, TOP_K, TOP_P):
        for ci, (vocab, temperature) in enumerate([(1000, 1.0), (37, 0.7), (5, 1.0), (4000, 0.0005), (64, 1.3)]):
            lg = rng.normal(0, 2.0, vocab).astype(np.float16)
            if ci == 2:
                lg[:] = lg[0]                    # all equal
            if ci == 4:
                lg[10:20] = np.float16(3.5)      # a plateau of equal best values
            cfg = {} if ci != 1 else {"max_k": 4, "top_p": 0.8, "pool_size": 20}
            out.append(dict(logits=lg, strategy=strategy, seed=1234 + ci, temperature=temperature, n_draws=24, text=[], config=cfg))
    for strategy in (MIN_P, TFS, TYPICAL, MIROSTAT):
        for ci, (vocab, temperature) in enumerate([(1000, 1.0), (60, 0.6), (3, 1.0), (4000, 1.0), (2, 0.6)]):
            lg = rng.normal(0, 2.5, vocab).astype(np.float16)
            if ci == 1:
                lg[5:9] = np.float16(4.0)
            out.append(dict(logits=lg, strategy=strategy, seed=99 + ci, temperature=temperature, n_draws=16, text=[], config={}))
    for strategy in (FSD, RANDOM_FSD):
        for ci, (vocab, ntext) in enumerate([(50, 40), (200, 3), (12, 1), (4000, 25)]):
            lg = rng.normal(0, 1.5, vocab).astype(np.float16)
            text = [int(t) for t in rng.integers(0, min(vocab, 8), ntext)]
            out.append(dict(logits=lg, strategy=strategy, seed=500 + ci, temperature=1.0, n_draws=16, text=text, config={}))
    return out


def main():
    if not os.path.exists(BIN):
        sys.exit("build oracle/_ref/ifa_ref_sampling first: make -C oracle ref_sampling (needs /root/reference)")
    arrays, meta = {}, []
    for i, c in enumerate(cases()):
        cfg = json.dumps(c["config"]) if c["config"] else ""
        ids, w, pid, pw = run_ref(c["logits"], c["strategy"], c["seed"], c["temperature"], c["n_draws"], c["text"], cfg)
        arrays["c%d_logits" % i] = c["logits"]
        arrays["c%d_text" % i] = np.asarray(c["text"], np.int32)
        arrays["c%d_ids" % i] = ids; arrays["c%d_w" % i] = w; arrays["c%d_pool_ids" % i] = pid; arrays["c%d_pool_w" % i] = pw
        meta.append(dict(strategy=c["strategy"], seed=c["seed"], temperature=c["temperature"], n_draws=c["n_draws"], config=c["config"]))
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_sampling.npz"), **arrays)
    print("wrote", len(meta), "cases")


if __name__ == "__main__":
    main()
