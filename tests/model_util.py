"""Build the oracle's whole-model restatement from the same synthetic tensors
the HIP worker was loaded with (tests only)."""
import numpy as np

import oracle as o
from inferflow_amd import dtypes as dt


def oracle_model_from_host(host, shape, max_ctx, kv_dtype=dt.F16, **cfg):
    m = o.Model(dim=shape["dim"], layers=shape["layers"], heads=shape["heads"], kv_heads=shape["kv_heads"],
                head_dim=shape["head_dim"], ffn=shape["ffn"], vocab=shape["vocab"], max_ctx=max_ctx,
                kv_dtype=kv_dtype, experts=shape.get("experts", 0), moe_top_k=shape.get("moe_top_k", 0), **cfg)
    for key, (target, arr, rows, cols) in host.items():
        layer, tid = key[0], key[1]
        expert = key[2] if len(key) > 2 else -1
        if target == dt.F16:
            data = arr.reshape(rows, cols).view(np.uint16)
        else:
            data = o.quantize(target, arr.reshape(rows, cols))
        m.set_tensor(max(layer, 0), tid, target, data, rows, cols, expert=expert)
    return m


# tensor ids (include/inferflow_amd.h)
_ROW_SPLIT = (12, 13, 14, 18, 20)          # wq, wk, wv, w1, w3: contiguous row ranges per rank (BY_COL of the reference)
_COL_SPLIT = (15, 19, 22, 23, 24, 26, 28)  # wo, w2: per-row byte slices (BY_ROW); the biases of the row-split matrices: column ranges of one row


def oracle_model_from_engine(eng, shape, max_ctx, kv_dtype=dt.F16, **cfg):
    """The whole model a (possibly partitioned) InferenceEngine holds, read BACK from its workers -- every rank's slice in
    reference-layout bytes, put together by the rules of SliceForWorker (host/model_loader.cc: row ranges for wq / wk / wv / w1 /
    w3 / lm_head, per-row column ranges for wo / w2, everything else replicated) -- and handed to the oracle.  Nothing is
    re-quantised on the host: the oracle multiplies exactly the blocks the device multiplies (the device quantiser itself is
    pinned bit for bit elsewhere: tests/test_gpu_ops.py)."""
    L, E = shape["layers"], shape.get("experts", 0)
    m = o.Model(dim=shape["dim"], layers=L, heads=shape["heads"], kv_heads=shape["kv_heads"], head_dim=shape["head_dim"], ffn=shape["ffn"],
                vocab=shape["vocab"], max_ctx=max_ctx, kv_dtype=kv_dtype, experts=E, moe_top_k=shape.get("moe_top_k", 0), **cfg)
    ranks = eng.model_info("partition_ranks")
    plans = [eng.worker_plan(r) for r in range(ranks)]

    def gather(layer, tid, expert=-1):
        """slices of the ranks that hold (global layer, tid), in tp_rank order"""
        parts = []
        for r, p in enumerate(plans):
            if layer >= 0:
                if not (p["layer0"] <= layer < p["layer1"]):
                    continue
                got = eng.worker_tensor(r, layer - p["layer0"], tid, expert)
            else:
                got = eng.worker_tensor(r, 0, tid)
            if got is not None:
                parts.append((p["tp_rank"], got))
        parts.sort(key=lambda x: x[0])
        return [g for _, g in parts]

    def put(layer, tid, expert=-1):
        parts = gather(layer, tid, expert)
        if not parts:
            return
        d = parts[0][0]
        if len(parts) == 1 or (tid not in _ROW_SPLIT and tid not in _COL_SPLIT and not (tid == 3 and layer < 0)):
            _, data, rows, cols = parts[0]                       # replicated (or a single rank)
            arr = data.view(np.uint16).reshape(rows, cols) if d == dt.F16 else data.reshape(rows, -1)
        elif tid in _COL_SPLIT:
            rows = parts[0][2]; cols = sum(p[3] for p in parts)
            arr = np.concatenate([p[1].reshape(rows, -1) for p in parts], axis=1)
            if d == dt.F16:
                arr = np.ascontiguousarray(arr).view(np.uint16).reshape(rows, cols)
        else:
            cols = parts[0][3]; rows = sum(p[2] for p in parts)
            arr = np.concatenate([p[1].reshape(p[2], -1) for p in parts], axis=0)
            if d == dt.F16:
                arr = np.ascontiguousarray(arr).view(np.uint16).reshape(rows, cols)
        m.set_tensor(max(layer, 0), tid, d, np.ascontiguousarray(arr), rows, cols, expert=expert)

    for tid in (0, 1, 2, 3):
        put(-1, tid)
    for layer in range(L):
        for tid in (10, 11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 25, 26, 27, 28):
            put(layer, tid)
        if E:
            for e in range(E):
                for tid in (18, 19, 20):
                    put(layer, tid, e)
        else:
            for tid in (18, 19, 20):
                put(layer, tid)
    return m


def oracle_model_from_worker(wk, shape, max_ctx, kv_dtype=dt.F16, **cfg):
    """The model ONE DecodeWorker holds, read back in reference-layout bytes (no host-side re-quantisation: both sides multiply the
    blocks the device quantiser wrote; the quantiser itself is pinned elsewhere), experts included."""
    L, E = shape["layers"], shape.get("experts", 0)
    m = o.Model(dim=shape["dim"], layers=L, heads=shape["heads"], kv_heads=shape["kv_heads"], head_dim=shape["head_dim"], ffn=shape["ffn"],
                vocab=shape["vocab"], max_ctx=max_ctx, kv_dtype=kv_dtype, experts=E, moe_top_k=shape.get("moe_top_k", 0), **cfg)

    def put(layer, tid, got, expert=-1):
        if got is None:
            return
        d, data, rows, cols = got
        m.set_tensor(max(layer, 0), tid, d, data.reshape(rows, cols) if d == dt.F16 else data.reshape(rows, -1), rows, cols, expert=expert)

    for tid in (0, 1, 2, 3):
        put(-1, tid, wk.get_tensor_host(0, tid))
    for layer in range(L):
        for tid in (10, 11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 25, 26, 27, 28) + (() if E else (18, 19, 20)):
            put(layer, tid, wk.get_tensor_host(layer, tid))
        for e in range(E):
            for tid in (18, 19, 20):
                put(layer, tid, wk.get_expert_tensor_host(layer, e, tid), e)
    return m
