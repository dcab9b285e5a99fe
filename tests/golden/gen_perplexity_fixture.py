#!/usr/bin/env python3
"""Golden perplexity numbers from the REFERENCE's own tool (src/tools/perplexity.cc compiled where it lies:
`make -C oracle ref_perplexity`) on a tiny llama2.c checkpoint -> tests/golden/ref_perplexity.npz.

The tool reads TEXT and tokenizes it with the model's vocabulary; the fixture's vocabulary gives every id >= 3 its own
single character, so that the text is a known id sequence (checked below: the PPL recomputed from the reference ENGINE's
logits on those ids -- oracle/_ref/ifa_ref_engine -- with the tool's formula must equal the tool's printed PPL).
What the fixture pins in host/perplexity.cc: BOS at the head of the stream, the max_length / stride windows, the
float log-softmax / double sums, the running and final estimates.

Needs /root/reference (build container only); the .npz travels."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import engine_fixtures as fx                     # noqa: E402
from tests.golden import gen_model_fixtures as gmf          # noqa: E402

TOOL = os.path.join(ROOT, "oracle", "_ref", "ifa_ref_perplexity")
SHAPE = dict(dim=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=1000)
SEED, STD, CTX = 29, 0.08, 128
N_TOKENS, MAX_LENGTH, STRIDE = 300, 96, 80


def char_of(i):
    return chr(0x4E00 + i)


def write_char_tokenizer(path, vocab):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 16))
        for i in range(vocab):
            s = ("<unk>", "<s>", "</s>")[i] if i < 3 else char_of(i)
            b = s.encode("utf-8")
            f.write(struct.pack("<fI", -float(i), len(b)))
            f.write(b)


PPL_INI = """[main]
inference_engine_config = {engine_ini}
test_data_file = {text}
max_length = {max_length}
stride = {stride}
temperature = 1.0

[app_env.base]
data_root_dir = {d}
require_enter_key_to_exit = false

[app_env.logging]
enable_logging = false
log_dir = {d}logs/
log_name = perplexity
color_console = false
"""


def tool_formula(windows_logits, windows_tokens):
    nll = nll2 = 0.0
    count, running = 0, []
    for lg, win in zip(windows_logits, windows_tokens):
        lg = lg.astype(np.float32)
        for i in range(len(win) - 1):
            row = lg[i]
            m = row.max()
            se = float(np.exp(row - m, dtype=np.float32).astype(np.float64).sum())
            v = -(float(row[win[i + 1]] - m) - np.log(se))
            nll += v; nll2 += v * v
        count += len(win) - 1
        running.append(float(np.exp(nll / count)))
    mean = nll / count
    var = nll2 / count - mean * mean
    ppl = float(np.exp(mean))
    return ppl, float(np.sqrt(var / (count - 1))) * ppl, count, running


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_perplexity", "ref_engine"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(SEED)
    body = rng.integers(3, SHAPE["vocab"], N_TOKENS).astype(np.int32)
    with tempfile.TemporaryDirectory() as d:
        d = d + "/"
        ini, _ = gmf.write_ref_model_dir(d, SHAPE, SEED, STD, CTX, False)
        write_char_tokenizer(os.path.join(d, "tokenizer.bin"), SHAPE["vocab"])
        text = os.path.join(d, "text.txt")
        open(text, "w", encoding="utf-8").write("".join(char_of(int(i)) for i in body) + "\n")
        ppl_ini = os.path.join(d, "perplexity.ini")
        open(ppl_ini, "w").write(PPL_INI.format(engine_ini=ini, text=text, max_length=MAX_LENGTH, stride=STRIDE, d=d))
        rel = os.path.relpath(ppl_ini, os.path.dirname(TOOL))
        r = subprocess.run([TOOL, rel], capture_output=True, text=True)
        out = r.stdout + r.stderr
        m = re.search(r"Final estimate: PPL = ([0-9.]+) \+/- ([0-9.]+)", out)
        if not m:
            raise RuntimeError("the reference tool printed no estimate:\n" + out[-3000:])
        ppl_tool, err_tool = float(m.group(1)), float(m.group(2))
        running_tool = [float(x) for x in re.findall(r"^\[\d+\]([0-9.]+)$", out, re.M)]
        # the id stream the tool scored: BOS (id 1) + the body, cut into windows like LoadQueryList
        ids = np.concatenate([[1], body]).astype(np.int32)
        wins = [ids[s:s + MAX_LENGTH] for s in range(0, len(ids), STRIDE)]
        logits = []
        for w in wins:
            rr = gmf.run_reference(ini, w, 1)
            logits.append(rr["prefill"])
    ppl, err, count, running = tool_formula(logits, wins)
    print("tool: PPL %.4f +/- %.5f, running %s" % (ppl_tool, err_tool, running_tool))
    print("recomputed from the reference engine's logits on the assumed ids: PPL %.4f +/- %.5f, running %s" % (ppl, err, ["%.4f" % x for x in running]))
    assert abs(ppl - ppl_tool) < 5e-4 and abs(err - err_tool) < 5e-5, "the text did not tokenize into the assumed ids"
    assert len(running_tool) == len(wins) and all(abs(a - b) < 5e-4 for a, b in zip(running, running_tool))
    path = os.path.join(ROOT, "tests", "golden", "ref_perplexity.npz")
    np.savez_compressed(path, shape=json.dumps(SHAPE), seed=SEED, std=STD, ctx=CTX, tokens=ids, max_length=MAX_LENGTH, stride=STRIDE,
                        ppl=ppl_tool, err=err_tool, running=np.array(running_tool), count=count)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
