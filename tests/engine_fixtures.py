"""Writers of tiny model directories for the InferenceEngine tests: a llama2.c checkpoint
(legacy layout, model_reader.cc:3248-3430), an HF-style safetensors checkpoint, and the .ini /
model_spec.json around them.  Weights are seeded N(0, 0.06) so greedy tokens vary."""
import json
import os
import struct

import numpy as np

from inferflow_amd import dtypes as dt
from inferflow_amd import worker as W

SHAPE = dict(dim=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=1000)
KINDS = [(W.T_ATTN_NORM, "norm"), (W.T_WQ, "q"), (W.T_WK, "kv"), (W.T_WV, "kv"), (W.T_WO, "o"),
         (W.T_FFN_NORM, "norm"), (W.T_W1, "up"), (W.T_W2, "down"), (W.T_W3, "up")]


def _shape(kind, s):
    d, f, kv = s["dim"], s["ffn"], s["kv_heads"] * s["head_dim"]
    return {"norm": (1, d), "q": (d, d), "kv": (kv, d), "o": (d, d), "up": (f, d), "down": (d, f)}[kind]


MOE_SHAPE = dict(dim=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=1000, experts=4, moe_top_k=2)


def make_weights(s=SHAPE, seed=5, std=0.06, shared_classifier=False):
    """{(layer, tid): float32 array [rows, cols]} incl. (-1, EMBD/OUT_NORM/LM_HEAD); mixture-of-experts shapes
    (s["experts"] > 0): the FFN matrices are (layer, tid, expert) and (layer, T_MOE_GATE) is the router."""
    rng = np.random.default_rng(seed)
    w = {(-1, W.T_EMBD): rng.normal(0, std, (s["vocab"], s["dim"])).astype(np.float32)}
    ne = s.get("experts", 0)
    for tid, kind in KINDS:
        for l in range(s["layers"]):
            r, c = _shape(kind, s)
            if ne and tid in (W.T_W1, W.T_W2, W.T_W3):
                for e in range(ne):
                    w[(l, tid, e)] = rng.normal(0, std, (r, c)).astype(np.float32)
                continue
            w[(l, tid)] = (1.0 + rng.normal(0, 0.02, (r, c))).astype(np.float32) if kind == "norm" else rng.normal(0, std, (r, c)).astype(np.float32)
    for l in range(s["layers"] if ne else 0):
        w[(l, W.T_MOE_GATE)] = rng.normal(0, 0.3, (ne, s["dim"])).astype(np.float32)
    w[(-1, W.T_OUT_NORM)] = (1.0 + rng.normal(0, 0.02, (1, s["dim"]))).astype(np.float32)
    w[(-1, W.T_LM_HEAD)] = w[(-1, W.T_EMBD)] if shared_classifier else rng.normal(0, std, (s["vocab"], s["dim"])).astype(np.float32)
    return w


def write_llama2c(path, w, s=SHAPE, seq_len=64, shared_classifier=False):
    with open(path, "wb") as f:
        f.write(struct.pack("<7i", s["dim"], s["ffn"], s["layers"], s["heads"], s["kv_heads"],
                            s["vocab"] if shared_classifier else -s["vocab"], seq_len))
        f.write(w[(-1, W.T_EMBD)].tobytes())
        for tid, _ in KINDS:
            for l in range(s["layers"]):
                f.write(w[(l, tid)].tobytes())
        f.write(w[(-1, W.T_OUT_NORM)].tobytes())
        # the reference skips seq_len*head_size BYTES here (model_reader.cc:3420); lay the file out accordingly
        f.write(b"\0" * (seq_len * s["head_dim"]))
        if not shared_classifier:
            f.write(w[(-1, W.T_LM_HEAD)].tobytes())


HF_NAMES = {W.T_ATTN_NORM: "input_layernorm.weight", W.T_WQ: "self_attn.q_proj.weight", W.T_WK: "self_attn.k_proj.weight",
            W.T_WV: "self_attn.v_proj.weight", W.T_WO: "self_attn.o_proj.weight", W.T_FFN_NORM: "post_attention_layernorm.weight",
            W.T_W1: "mlp.gate_proj.weight", W.T_W2: "mlp.down_proj.weight", W.T_W3: "mlp.up_proj.weight"}


def fuse_qkv(w, l, s, fmt):
    """[q; k; v] rows in the two layouts of ModelSpec::qkv_format (0: per KV group {q heads, k, v}; 1: q | k | v)."""
    q, k, v = w[(l, W.T_WQ)], w[(l, W.T_WK)], w[(l, W.T_WV)]
    if fmt == 1:
        return np.concatenate([q, k, v], 0)
    hd, groups = s["head_dim"], s["kv_heads"]
    hq = s["heads"] // groups
    parts = []
    for g in range(groups):
        parts += [q[g * hq * hd:(g + 1) * hq * hd], k[g * hd:(g + 1) * hd], v[g * hd:(g + 1) * hd]]
    return np.concatenate(parts, 0)


def write_safetensors(path, w, s=SHAPE, dtype="F16", fused_qkv=None):
    """Minimal safetensors writer (8-byte header length, JSON header, raw little-endian payload).
    fused_qkv: None, or the qkv_format (0 / 1) of a single self_attn.qkv_proj tensor replacing q/k/v_proj."""
    tensors = {"model.embed_tokens.weight": w[(-1, W.T_EMBD)], "model.norm.weight": w[(-1, W.T_OUT_NORM)].reshape(-1),
               "lm_head.weight": w[(-1, W.T_LM_HEAD)]}
    for key, arr in w.items():
        l, tid = key[0], key[1]
        if len(key) == 3:        # Mixtral names (data/models/mixtral_8x7b_instruct_v0.1/model_spec.safetensors.json)
            tensors["model.layers.%d.block_sparse_moe.experts.%d.%s.weight" % (l, key[2], {W.T_W1: "w1", W.T_W2: "w2", W.T_W3: "w3"}[tid])] = arr
            continue
        if l >= 0 and tid == W.T_MOE_GATE:
            tensors["model.layers.%d.block_sparse_moe.gate.weight" % l] = arr
            continue
        if l >= 0:
            if fused_qkv is not None and tid in (W.T_WQ, W.T_WK, W.T_WV):
                if tid == W.T_WQ:
                    tensors["model.layers.%d.self_attn.qkv_proj.weight" % l] = fuse_qkv(w, l, s, fused_qkv)
                continue
            tensors["model.layers.%d.%s" % (l, HF_NAMES[tid])] = arr.reshape(-1) if arr.shape[0] == 1 else arr
    header, blobs, off = {}, [], 0
    for name, arr in tensors.items():
        if dtype == "F16":
            raw = arr.astype(np.float16).tobytes()
        elif dtype == "BF16":
            raw = (arr.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16).tobytes()   # truncation: exact for the test's values? no -> see f16 rounding below
        else:
            raw = arr.astype(np.float32).tobytes()
        header[name] = {"dtype": dtype, "shape": list(arr.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw); off += len(raw)
    hj = json.dumps(header).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj))); f.write(hj)
        for b in blobs:
            f.write(b)


SPEC = {
    "config_file": "", "model_files": ["model.bin"], "model_file_format": "llama2.c",
    "tokenizer_file": "", "tokenization_algorithm": "bpe", "generation_config": "",
    "network_structure": {"type": "transformer.llama", "normalization_function": "rms", "activation_function": "silu",
                          "position_embedding": "rope", "qk_column_order": 0, "tensor_name_prefix": "", "tensor_name_mapping": {}},
}

INI = """; written by tests/engine_fixtures.py in the reference's llm_inference.ini dialect
[main]
inference_engine_config = ${{config_dir}}/engine.ini

[transformer_engine]
models = {name}
devices = {devices}
force_partition_path = {force_partition}
decoder_cpu_layer_count = 0
cpu_threads = 8
max_concurrent_queries = {maxq}
return_output_tensors = {ret}
dynamic_batching_min_queries = 2

[model.{name}]
model_dir = ${{config_dir}}
model_specification_file = model_spec.json
device_weight_data_type = {wd}
device_kv_cache_data_type = {kvd}
tensor_quant_threshold = {thr}
max_context_len = {ctx}
prompt_template = {{bos}}{{query}}
"""


def write_model_dir(d, fmt="llama2.c", wd="Q4", kvd="Q8", thr=0, ctx=64, ret="true", maxq=6, s=SHAPE, seed=5, qk_order=0,
                    st_dtype="F16", hyper=None, fused_qkv=None, std=0.06, shared_classifier=False, devices="0", force_partition="false", net_extra=None):
    """Returns (ini_path, weights dict or None).  net_extra: more network_structure keys (scales, linear norm ...)."""
    os.makedirs(d, exist_ok=True)
    spec = json.loads(json.dumps(SPEC))
    spec["network_structure"]["qk_column_order"] = qk_order
    spec["network_structure"].update(net_extra or {})
    w = None
    if fmt == "llama2.c":
        w = make_weights(s, seed, std, shared_classifier=shared_classifier)
        write_llama2c(os.path.join(d, "model.bin"), w, s, seq_len=ctx, shared_classifier=shared_classifier)
    elif fmt == "safetensors":
        w = make_weights(s, seed, std)
        spec.update(model_file_format="safetensors", model_files=["model.safetensors.index.json", "model.safetensors"], config_file="config.json")
        spec["network_structure"]["tensor_name_prefix"] = "model."
        write_safetensors(os.path.join(d, "model.safetensors"), w, s, st_dtype, fused_qkv)
        if fused_qkv is not None:
            spec["network_structure"]["qkv_format"] = fused_qkv
        cfgj = {"hidden_size": s["dim"], "intermediate_size": s["ffn"], "num_hidden_layers": s["layers"],
                "num_attention_heads": s["heads"], "num_key_value_heads": s["kv_heads"], "vocab_size": s["vocab"],
                "max_position_embeddings": 2048, "rope_theta": 10000.0}
        if s.get("experts", 0):
            spec["network_structure"].update(type="transformer.decoder_only.sparse_moe", expert_count=s["experts"], using_expert_count=s["experts"],
                                             moe_top_k=s["moe_top_k"])
            cfgj.update(num_local_experts=s["experts"], num_experts_per_tok=s["moe_top_k"])
        json.dump(cfgj, open(os.path.join(d, "config.json"), "w"))
    else:
        spec.update(model_file_format="synthetic", model_files=[])
        spec["hyper_params"] = hyper or {"vocab_size": s["vocab"], "embd_dims": s["dim"], "hidden_dim": s["ffn"], "decoder_layers": s["layers"],
                                         "decoder_heads": s["heads"], "decoder_kv_heads": s["kv_heads"]}
        if s.get("experts", 0):
            spec["network_structure"].update(type="transformer.decoder_only.sparse_moe", expert_count=s["experts"], moe_top_k=s["moe_top_k"])
        spec["synthetic_std"] = 0.06
    json.dump(spec, open(os.path.join(d, "model_spec.json"), "w"), indent=2)
    ini = os.path.join(d, "engine.ini")
    open(ini, "w").write(INI.format(name="tiny_test", wd=wd, kvd=kvd, thr=thr, ctx=ctx, ret=ret, maxq=maxq, devices=devices,
                                   force_partition=force_partition))
    return ini, w


def host_tensors(w, s, wdtype, thr=0, lm_quant=True):
    """The oracle's view of what the engine loads: F16-rounded sources + target dtypes (NetworkBuilder policy)."""
    host = {}
    for key, arr in w.items():
        l, tid = key[0], key[1]
        rows, cols = arr.shape
        is_matrix = tid in (W.T_WQ, W.T_WK, W.T_WV, W.T_WO, W.T_W1, W.T_W2, W.T_W3)
        target = dt.F16
        if is_matrix and wdtype >= 7 and rows * cols >= thr and cols % dt.block_capacity(wdtype) == 0:
            target = wdtype
        if tid == W.T_LM_HEAD and l < 0 and wdtype >= 7 and s["layers"] <= 20 and lm_quant:
            target = wdtype
        host[key] = (target, arr.astype(np.float16), rows, cols)
    return host
