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
