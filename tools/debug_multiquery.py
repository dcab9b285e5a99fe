#!/usr/bin/env python3
"""Diagnosis aid (GPU box): several queries registered BEFORE the first Infer() on an .ini engine (devices = 0 | 0&0), each checked
against the oracle run on that query alone: prompt rows, then batched decode steps.
    python tools/debug_multiquery.py <shape> <layers> <nq>"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from inferflow_amd import dtypes as dt, synth
from inferflow_amd.engine import InferenceEngine
from tests.model_util import oracle_model_from_engine


def cm(a, b):
    a = a.astype(np.float32); b = b.astype(np.float32)
    return float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)), float(np.abs(a - b).max() / b.std())


def main():
    shape, layers, nq = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    s = dict(synth.SHAPES[shape]); s["layers"] = layers
    E = s.get("experts", 0)
    for devices, merge in (("0", 1), ("0&0", 2)):
        d = tempfile.mkdtemp(prefix="ifa_dbg_")
        hp = {"vocab_size": s["vocab"], "embd_dims": s["dim"], "hidden_dim": s["ffn"], "decoder_layers": layers, "decoder_heads": s["heads"], "decoder_kv_heads": s["kv_heads"]}
        ns = {"type": "transformer.llama", "normalization_function": "rms", "activation_function": "silu", "position_embedding": "rope", "qk_column_order": 2,
              "tensor_name_prefix": "", "tensor_name_mapping": {}}
        if E:
            ns.update({"expert_count": E, "moe_top_k": s["moe_top_k"]})
        json.dump({"config_file": "", "model_files": [], "model_file_format": "synthetic", "tokenizer_file": "", "tokenization_algorithm": "bpe", "generation_config": "",
                   "synthetic_std": 0.02, "hyper_params": hp, "network_structure": ns}, open(os.path.join(d, "model_spec.json"), "w"))
        ini = os.path.join(d, "engine.ini")
        open(ini, "w").write("[transformer_engine]\nmodels = m\ndevices = %s\ndecoder_cpu_layer_count = 0\ncpu_threads = 8\nmax_concurrent_queries = %d\n"
                             "return_output_tensors = true\n\n[model.m]\nmodel_dir = ${config_dir}\nmodel_specification_file = model_spec.json\n"
                             "device_weight_data_type = Q4\ndevice_kv_cache_data_type = F16\ntensor_quant_threshold = 0\nmax_context_len = 64\nprompt_template = {bos}{query}\n" % (devices, nq))
        eng = InferenceEngine.from_ini(ini)
        om = oracle_model_from_engine(eng, s, 64, dt.F16, unk_id=0, tp_merge=merge)
        rng = np.random.default_rng(87)
        prompts = [rng.integers(3, s["vocab"], 2 + i % 3).astype(np.int32) for i in range(nq)]
        qids = [eng.add_query(p) for p in prompts]
        first = dict(eng.infer())
        rows = {q: eng.last_logits(q).copy() for q in qids}
        orc = {}
        for qi, q in enumerate(qids):
            om.reset()
            t, lg = om.forward(prompts[qi], 0)
            print("devices=%s query %d (slot %d, prompt %d): prompt rows %s" % (devices, q, qi, len(prompts[qi]),
                  " ".join("%.5f/%.3f" % cm(rows[q][i], lg[i]) for i in range(len(prompts[qi])))), flush=True)
            cur, pos, rr, tt = int(t), len(prompts[qi]), [], [int(t)]
            for _ in range(3):
                t, l1 = om.forward(np.array([cur], np.int32), pos)
                rr.append(l1[0].copy()); tt.append(int(t)); cur, pos = int(t), pos + 1
            orc[q] = (rr, tt)
        for step in range(3):
            eng.commit({q: orc[q][1][step] for q in qids})
            got = dict(eng.infer())
            print("devices=%s batched step %d: %s" % (devices, step, " ".join("%.5f/%.3f" % cm(eng.last_logits(q)[0], orc[q][0][step]) for q in qids)), flush=True)
        eng.close(); del om


if __name__ == "__main__":
    main()
