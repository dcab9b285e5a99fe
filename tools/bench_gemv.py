#!/usr/bin/env python3
"""Micro-benchmark of the GEMV kernels at Llama-2-7B shapes (GPU box only).
Rotates over enough distinct weight copies to defeat the 256 MiB Infinity Cache.
Prints one JSON line per variant: achieved algorithmic GB/s."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import inferflow_amd as ia  # noqa: E402
from inferflow_amd import dtypes as dt  # noqa: E402
from tests import gpu_util as g  # noqa: E402


def timeit(fn, n_iter, warmup=3):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n_iter):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n_iter


def main():
    L = ia.lib()
    res = []
    d = dt.Q4_B32T1A
    for rows, cols in [(4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008)]:
        nbytes = rows * dt.row_bytes(d, cols)
        ncopy = max(2, int(400e6 // nbytes) + 1)
        w = (torch.randn(rows, cols, device="cuda") * 0.02).half()
        Wq = g.quantize(d, w)
        Wt = g.repack(d, Wq, rows, cols)
        Ws = [Wq.clone() for _ in range(ncopy)]
        Wts = [Wt.clone() for _ in range(ncopy)]
        x = torch.randn(1, cols, device="cuda").half()
        xq = g.quantize_act(x)
        y = g.empty_f16(rows)
        st = g.stream()

        def aos(i):
            ia.check(L.ifa_gemv(d, g.p(Ws[i % ncopy]), rows, cols, dt.Q8_B32T2, g.p(xq), None, g.p(y), st))

        def tiled(i):
            ia.check(L.ifa_gemv_tiled(d, g.p(Wts[i % ncopy]), rows, cols, g.p(xq), None, g.p(y), st))

        for name, fn in (("q4_aos", aos), ("q4_tiled", tiled)):
            t = timeit(fn, 200)
            res.append({"kernel": name, "rows": rows, "cols": cols, "us": t * 1e6, "GBps": nbytes / t / 1e9})
            print(json.dumps(res[-1]), flush=True)
        del Ws, Wts
    # F16 lm_head
    rows, cols = 32000, 4096
    Ws = [(torch.randn(rows, cols, device="cuda") * 0.02).half() for _ in range(3)]
    x = torch.randn(cols, device="cuda").half()
    y = g.empty_f16(rows)
    st = g.stream()

    def f16(i):
        ia.check(L.ifa_gemv(dt.F16, g.p(Ws[i % 3]), rows, cols, dt.F16, g.p(x), None, g.p(y), st))
    t = timeit(f16, 100)
    print(json.dumps({"kernel": "f16_lm_head", "rows": rows, "cols": cols, "us": t * 1e6,
                      "GBps": rows * cols * 2 / t / 1e9}), flush=True)
    # device copy ceiling (read+write 1 GiB)
    a = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    t = timeit(lambda i: b.copy_(a), 20)
    print(json.dumps({"kernel": "torch_copy_1GiB", "GBps_rw": 2 * a.numel() * 4 / t / 1e9}), flush=True)
    t = timeit(lambda i: a.sum(), 20)
    print(json.dumps({"kernel": "torch_sum_1GiB_read", "GBps": a.numel() * 4 / t / 1e9}), flush=True)


if __name__ == "__main__":
    main()
