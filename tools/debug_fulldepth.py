#!/usr/bin/env python3
"""Diagnosis aid (GPU box): MoE / GQA models at FULL WIDTH but few layers against the oracle, on one worker and through the
.ini engine with devices = 0&0, for prompts of 1..5 tokens -- which shape / path / prompt length parts from the oracle.
    python tools/debug_fulldepth.py [mixtral_8x7b|yi_34b] [layers]"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle as o
from inferflow_amd import dtypes as dt, synth
from inferflow_amd.engine import InferenceEngine
from tests import gpu_util as g
from tests.model_util import oracle_model_from_engine


def cm(a, b):
    a = a.astype(np.float32); b = b.astype(np.float32)
    return float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)), float(np.abs(a - b).max() / b.std())


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "mixtral_8x7b"
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    s0 = dict(synth.SHAPES[shape]); s0["layers"] = layers
    # ---- one worker
    wk, _, s = synth.build(shape, dt.Q4_B32T1A, dt.F16, max_ctx=64, layers=layers)
    om = o.Model(dim=s["dim"], layers=layers, heads=s["heads"], kv_heads=s["kv_heads"], head_dim=s["head_dim"], ffn=s["ffn"], vocab=s["vocab"],
                 max_ctx=64, kv_dtype=dt.F16, experts=s.get("experts", 0), moe_top_k=s.get("moe_top_k", 0))
    E = s.get("experts", 0)
    for layer in range(-1, layers):
        for tid in ((0, 1, 2, 3) if layer < 0 else (10, 11, 12, 13, 14, 15, 16, 17, 21) + (() if E else (18, 19, 20))):
            got = wk.get_tensor_host(max(layer, 0), tid)
            if got is None:
                continue
            d, data, rows, cols = got
            om.set_tensor(max(layer, 0), tid, d, data.reshape(rows, -1) if d != dt.F16 else data.reshape(rows, cols), rows, cols)
    if E:
        import ctypes as C
        from inferflow_amd._capi import lib, check
        for layer in range(layers):
            for e in range(E):
                for tid in (18, 19, 20):
                    d, p, r, c = C.c_int(), C.c_void_p(), C.c_size_t(), C.c_size_t()
                    check(lib().ifa_model_get_expert_tensor(wk._h, layer, e, tid, C.byref(d), C.byref(p), C.byref(r), C.byref(c)))
                    nb = r.value * dt.row_bytes(d.value, c.value)
                    out = np.empty(nb, np.uint8)
                    check(lib().ifa_memcpy_d2h(out.ctypes.data_as(C.c_void_p), p, nb, None)); check(lib().ifa_stream_sync(None))
                    om.set_tensor(layer, tid, d.value, out.reshape(r.value, -1), r.value, c.value, expert=e)
    rng = np.random.default_rng(87)
    for T in (1, 2, 3, 4, 5, 8):
        pr = rng.integers(3, s["vocab"], T).astype(np.int32)
        om.reset(); t_o, lg_o = om.forward(pr, 0)
        for opts in ({}, {"moe_device": 0}, {"rows_mo": 0}, {"moe_singles": 0}, {"batch_fused": 0}):
            for k, v in opts.items():
                wk.set_option(k, v)
            wk.reset()
            lg = torch.empty((T, s["vocab"]), dtype=torch.float16, device="cuda")
            try:
                t_g = wk.forward(pr, 0, lg)
                res = " ".join("%.5f/%.3f" % cm(g.host(lg)[i], lg_o[i]) for i in range(T))
            except Exception as ex:      # noqa: BLE001
                res = "FAILED %r" % (ex,)
            for k in opts:
                wk.set_option(k, 1)
            print("worker T=%d %-18s rows cos/|d|std: %s" % (T, opts or "default", res), flush=True)
        # the same prompt token by token through the decode path
        wk.reset(); om.reset()
        outs = []
        for i, t in enumerate(pr):
            wk.decode(int(t), i, 1)
            lg1 = wk.read_buffer("logits").view(np.float16).copy()
            _, l1 = om.forward(np.array([t], np.int32), i)
            outs.append("%.5f/%.3f" % cm(lg1, l1[0]))
        print("worker T=%d token-by-token decode path: %s" % (T, " ".join(outs)), flush=True)
    wk.close(); del om
    # ---- the .ini engine, devices = 0&0
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
        open(ini, "w").write("[transformer_engine]\nmodels = m\ndevices = %s\ndecoder_cpu_layer_count = 0\ncpu_threads = 8\nmax_concurrent_queries = 8\n"
                             "return_output_tensors = true\n\n[model.m]\nmodel_dir = ${config_dir}\nmodel_specification_file = model_spec.json\n"
                             "device_weight_data_type = Q4\ndevice_kv_cache_data_type = F16\nmax_context_len = 64\nprompt_template = {bos}{query}\n" % devices)
        eng = InferenceEngine.from_ini(ini)
        om = oracle_model_from_engine(eng, s0, 64, dt.F16, unk_id=0, tp_merge=merge)
        for T in (1, 2, 3, 4, 5):
            pr = rng.integers(3, s["vocab"], T).astype(np.int32)
            om.reset(); t_o, lg_o = om.forward(pr, 0)
            q = eng.add_query(pr)
            (qq, tok), = eng.infer()
            lg = eng.last_logits(q)
            res = " ".join("%.5f/%.3f" % cm(lg[i], lg_o[i]) for i in range(lg.shape[0]))
            # two decode steps behind it
            cur, pos, dec = int(t_o), T, []
            for _ in range(2):
                t2, l2 = om.forward(np.array([cur], np.int32), pos)
                eng.commit({q: cur}); (qq, tk), = eng.infer()
                dec.append("%.5f/%.3f" % cm(eng.last_logits(q)[0], l2[0]))
                cur, pos = int(t2), pos + 1
            eng.remove_query(q)
            print("engine devices=%s T=%d prompt rows: %s | decode: %s" % (devices, T, res, " ".join(dec)), flush=True)
        eng.close(); del om


if __name__ == "__main__":
    main()
