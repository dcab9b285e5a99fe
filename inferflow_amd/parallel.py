"""Mapping of the model onto the GPUs of one node (one process per GPU).

world == 1: a single DecodeWorker.
world  > 1: tensor-parallel group, see TPRunner (BY_TENSOR partition of the
reference, src/transformer/network_builder.cc:1594-1686).
"""
import numpy as np

from . import dtypes as dt, synth, worker as W

ALL_TIDS = [W.T_EMBD, W.T_OUT_NORM, W.T_OUT_NORM_B, W.T_LM_HEAD] + list(range(10, 29))


class SingleRunner:
    def __init__(self, shape_name, wdtype, kv_dtype, max_ctx, device=0):
        self.worker, _, self.shape = synth.build(shape_name, wdtype, kv_dtype, max_ctx=max_ctx, device=device)
        ok, why = self.worker.fused_supported()
        if not ok:
            raise RuntimeError("fused decode path unavailable: " + why)

    def prefill(self, prompt):
        return self.worker.forward(np.asarray(prompt, np.int32), 0)

    def decode_prepare(self, pos, n):
        """Capture what decode(., pos, n) replays (no step runs): keeps graph capture out of a caller's timed region."""
        self.worker.decode_prepare(pos, min(n, 1024))

    def decode(self, tok, pos, n):
        out, ms_total = [], 0.0
        while n > 0:                      # the device token ring holds 1024 steps per call
            k = min(n, 1024)
            t, ms = self.worker.decode(tok, pos, k)
            out.extend(int(x) for x in t)
            ms_total += ms
            tok, pos, n = int(t[-1]), pos + k, n - k
        return out, ms_total

    def export_host_tensors(self):
        host = {}
        for tid in ALL_TIDS:
            layers = [-1] if tid < 10 else range(self.shape["layers"])
            for layer in layers:
                t = self.worker.get_tensor_host(max(layer, 0), tid)
                if t is not None:
                    host[(layer, tid)] = t
        return host


def build_runner(shape_name, wdtype, kv_dtype, max_ctx, world=1, rank=0, local_rank=0, groups=1):
    """groups = number of device groups (layer ranges); world // groups ranks per group are tensor-parallel."""
    import os
    if world == 1 and not os.environ.get("IFA_FORCE_TP"):
        return SingleRunner(shape_name, wdtype, kv_dtype, max_ctx, device=local_rank)
    from .tp import CTPRunner, TPRunner
    # the C path (collectives of the C ABI, the step driven from C) unless IFA_TP_BACKEND=torch asks for the round-1
    # runner (torch.distributed collectives around the worker segments)
    if os.environ.get("IFA_TP_BACKEND", "c") != "torch":
        return CTPRunner(shape_name, wdtype, kv_dtype, max_ctx, world, rank, local_rank, groups=groups)
    return TPRunner(shape_name, wdtype, kv_dtype, max_ctx, world, rank, local_rank, groups=groups)
