"""Generator of the whole-model fixtures tests/golden/ref_model_*.npz (SURVEY.md 8c "F3").

Runs in the BUILD container only (needs /root/reference): `make -C oracle ref_engine` compiles the reference's own CPU
inference path (InferenceEngine + CpuInferenceWorker + ggml, sources where they lie, plain g++/gcc) with
oracle/ref_engine_driver.cc on top; this script writes seeded synthetic llama2.c checkpoints (the layout of
src/transformer/model_reader.cc:3248-3430, a tokenizer file in the format of ReadVocabulary_Format2 :1362-1417, the
.ini / model_spec.json the reference reads), runs the reference on them with return_output_tensors = true and stores

    shape, seed, std          -> how tests regenerate the SAME checkpoint (tests/engine_fixtures.make_weights)
    prompt                    int32 [P]
    tokens                    int32 [N]     greedy ids of N steps (step 0 = prefill)
    prefill_logits            float16 [P][vocab]   (the reference's values are fp32; fp16 keeps the fixture small and is
                                                    far inside the 1e-2 comparison tolerance)
    step_logits               float16 [N-1][vocab]
    top2_gap                  float32 [N]   top-1 minus top-2 of the reference's fp32 logits over the allowed ids
    excluded_ids              int32 []      ids GetSortedTopK never offers (the unk id; sampling_strategy.cc:281-297)

The `st_*` cases are the same kind of run on a SAFETENSORS directory (HF tensor names under the "model." prefix, config.json
hyper-parameters, qk_column_order 2 = HF's rotate-half column order, a vocab.txt read by LoadTokenizer_Txt
model_reader.cc:1098-1137): they pin the safetensors loader's name map and the RoPE pairing of qk_column_order 2 to the
reference engine itself (fields fmt / qk_order say how tests must write the model directory).

Nothing here is imported by the product or at GPU-test time; the .npz files are data.
    python tests/golden/gen_model_fixtures.py
"""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import engine_fixtures as fx  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ifa_ref_engine")

# name -> (shape, seed, std, prompt_len, steps, shared_classifier)
CASES = {
    "mha": (dict(dim=256, layers=2, heads=4, kv_heads=4, head_dim=64, ffn=512, vocab=1000), 21, 0.06, 9, 72, False),
    "gqa": (dict(dim=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=1000), 5, 0.06, 9, 72, False),
    "gqa_deep": (dict(dim=384, layers=4, heads=6, kv_heads=2, head_dim=64, ffn=1024, vocab=1200), 33, 0.05, 17, 72, True),
    "stories15m_shape": (dict(dim=288, layers=6, heads=6, kv_heads=6, head_dim=48, ffn=768, vocab=2000), 77, 0.05, 12, 72, True),
}
# safetensors directories: name -> (shape, seed, std, prompt_len, steps, qk_column_order)
ST_CASES = {
    "st_gqa_hf": (dict(dim=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=1000), 57, 0.06, 11, 72, 2),
    "st_mha_interleaved": (dict(dim=256, layers=2, heads=4, kv_heads=4, head_dim=64, ffn=512, vocab=1000), 43, 0.06, 9, 64, 0),
}

REF_INI = """[transformer_engine]
models = tiny_ref
devices = 0
decoder_cpu_layer_count = 1000
cpu_threads = 4
max_concurrent_queries = 2
return_output_tensors = true
is_study_mode = false
show_tensors = false

[model.tiny_ref]
model_dir = ${{config_dir}}
model_specification_file = model_spec.json
device_weight_data_type = F16
device_kv_cache_data_type = F16
host_kv_cache_percent = 0
max_context_len = {ctx}
decoding_strategy = greedy
prompt_template = {{bos}}{{query}}

[prompt_templates]
prompt_template_count = 0

[app_env.base]
data_root_dir = ${{config_dir}}
require_enter_key_to_exit = false

[app_env.logging]
enable_logging = 0
log_dir = ${{config_dir}}logs
log_name = ref

[app_env.status_manager]
enable_monitoring = 0
"""


def write_tokenizer(path, vocab):
    """llama2.c tokenizer.bin as ReadVocabulary_Format2 reads it: u32 max_token_len, then per token f32 score, u32 len, bytes."""
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 16))
        for i in range(vocab):
            s = ("<unk>", "<s>", "</s>")[i] if i < 3 else "t%d" % i
            b = s.encode()
            f.write(struct.pack("<fI", -float(i), len(b)))
            f.write(b)


def write_ref_model_dir(d, shape, seed, std, ctx, shared):
    os.makedirs(d, exist_ok=True)
    w = fx.make_weights(shape, seed, std, shared_classifier=shared)
    fx.write_llama2c(os.path.join(d, "model.bin"), w, shape, seq_len=ctx, shared_classifier=shared)
    write_tokenizer(os.path.join(d, "tokenizer.bin"), shape["vocab"])
    spec = json.loads(json.dumps(fx.SPEC))
    spec["tokenizer_file"] = "tokenizer.bin"
    spec["qkv_format"] = 1
    json.dump(spec, open(os.path.join(d, "model_spec.json"), "w"), indent=2)
    ini = os.path.join(d, "engine.ini")
    open(ini, "w").write(REF_INI.format(ctx=ctx))
    return ini, w


def write_ref_safetensors_dir(d, shape, seed, std, ctx, qk_order):
    """HF-named F16 safetensors + config.json + vocab.txt, the spec keys of data/models/*/model_spec.safetensors.json."""
    os.makedirs(d, exist_ok=True)
    w = fx.make_weights(shape, seed, std)
    fx.write_safetensors(os.path.join(d, "model.safetensors"), w, shape, "F16")
    with open(os.path.join(d, "vocab.txt"), "w") as f:
        for i in range(shape["vocab"]):
            f.write(("<unk>", "<s>", "</s>")[i] + "\n" if i < 3 else "t%d\n" % i)
    spec = json.loads(json.dumps(fx.SPEC))
    spec.update(model_file_format="safetensors", model_files=["model.safetensors"], config_file="config.json", tokenizer_file="vocab.txt")
    ns = spec["network_structure"]
    ns["tensor_name_prefix"] = "model."
    ns["qk_column_order"] = qk_order
    json.dump(spec, open(os.path.join(d, "model_spec.json"), "w"), indent=2)
    json.dump({"hidden_size": shape["dim"], "intermediate_size": shape["ffn"], "num_hidden_layers": shape["layers"],
               "num_attention_heads": shape["heads"], "num_key_value_heads": shape["kv_heads"], "vocab_size": shape["vocab"],
               "max_position_embeddings": 2048, "rope_theta": 10000.0}, open(os.path.join(d, "config.json"), "w"))
    ini = os.path.join(d, "engine.ini")
    open(ini, "w").write(REF_INI.format(ctx=ctx))
    return ini, w


def run_reference(ini, prompt, steps, quiet=False):
    d = os.path.dirname(ini)
    pf = os.path.join(d, "prompt.i32")
    np.asarray(prompt, np.int32).tofile(pf)
    out = os.path.join(d, "out.bin")
    cmd = [DRIVER, ini, pf, str(steps), out] + (["quiet"] if quiet else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference driver failed (%d):\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-2000:]))
    raw = open(out, "rb").read()
    magic, vocab, plen, n = struct.unpack_from("<4i", raw, 0)
    assert magic == 0x49464131 and plen == len(prompt) and n == steps
    off = 16
    toks, step_logits, prefill = [], [], None
    for s in range(steps):
        if not quiet:
            rows = plen if s == 0 else 1
            lg = np.frombuffer(raw, np.float32, rows * vocab, off).reshape(rows, vocab); off += rows * vocab * 4
            if s == 0:
                prefill = lg
            else:
                step_logits.append(lg[0])
        toks.append(struct.unpack_from("<i", raw, off)[0]); off += 4
    prefill_ms, decode_ms = struct.unpack_from("<2d", raw, off)
    return dict(tokens=np.array(toks, np.int32), prefill=prefill, steps=np.array(step_logits, np.float32) if step_logits else None,
                prefill_ms=prefill_ms, decode_ms=decode_ms, log=r.stderr)


def main():
    if not os.path.exists(DRIVER):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_engine"])
    cases = [(n, c[:5] + (c[5], "llama2.c", 0)) for n, c in CASES.items()] + [(n, c[:5] + (False, "safetensors", c[5])) for n, c in ST_CASES.items()]
    for name, (shape, seed, std, plen, steps, shared, fmt, qk_order) in cases:
        with tempfile.TemporaryDirectory() as d:
            ctx = 128
            if fmt == "safetensors":
                ini, _ = write_ref_safetensors_dir(d + "/", shape, seed, std, ctx, qk_order)
            else:
                ini, _ = write_ref_model_dir(d + "/", shape, seed, std, ctx, shared)
            prompt = np.random.default_rng(1000 + seed).integers(3, shape["vocab"], plen).astype(np.int32)
            r = run_reference(ini, prompt, steps)
        allrows = np.concatenate([r["prefill"][-1:], r["steps"]], 0)
        # GetSortedTopK (sampling_strategy.cc:281-297) never offers the unk id (0 here) or Invalid-type tokens to the
        # queue: the greedy id is the argmax over the ALLOWED ids
        allowed = allrows.copy()
        allowed[:, 0] = -np.inf
        srt = np.sort(allowed, axis=1)
        gap = (srt[:, -1] - srt[:, -2]).astype(np.float32)
        assert (np.argmax(allowed, 1) == r["tokens"]).all(), "reference greedy ids are not the argmax of its logits"
        path = os.path.join(ROOT, "tests", "golden", "ref_model_%s.npz" % name)
        np.savez_compressed(path, shape=json.dumps(shape), seed=seed, std=std, shared_classifier=shared, ctx=ctx, prompt=prompt,
                            tokens=r["tokens"], prefill_logits=r["prefill"].astype(np.float16),
                            step_logits=r["steps"].astype(np.float16), top2_gap=gap, excluded_ids=np.array([0], np.int32),
                            fmt=fmt, qk_order=qk_order)
        print("%s: %d distinct greedy ids in %d steps, min top-2 gap %.4f, %d KB" % (
            name, len(set(r["tokens"].tolist())), steps, gap.min(), os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
