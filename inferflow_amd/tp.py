"""Tensor-parallel mapping of the decode path onto the GPUs of one node
(one process per GPU, torch.distributed: backend "nccl" == RCCL over xGMI).

Partition = the reference's BY_TENSOR strategy
(src/transformer/network_builder.cc:1594-1686, device_tensor_builder.cu:203-239):
  wq / wk / wv / w1 / w3 : contiguous ROW ranges  (heads, kv heads and FFN rows split)
  wo / w2                : contiguous COLUMN ranges (whole quant blocks: cols/p % 32 == 0)
  norms, embeddings      : replicated
  KV cache               : sharded by KV head
  lm_head                : sharded by vocabulary rows (the reference keeps it on the last
                           rank only; sharding it and doing a distributed argmax is the
                           extension SURVEY.md §8e names)
Exchange = two sum all-reduces of one [dim] F16 vector per layer (after wo, after w2),
exactly where the reference calls DistributeAndMergeTensors.  At batch-1 decode the
payload is 8 KB, i.e. latency-bound.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import dtypes as dt, synth, worker as W

ROW_SPLIT = {W.T_WQ, W.T_WK, W.T_WV, W.T_W1, W.T_W3}
COL_SPLIT = {W.T_WO, W.T_W2}


def check_divisible(shape, world):
    """Same constraints as the reference (network_builder.cc:1207-1213) + whole quant blocks."""
    if shape["heads"] % world or shape["kv_heads"] % world:
        raise ValueError("heads (%d) and kv_heads (%d) must be divisible by the group size %d"
                         % (shape["heads"], shape["kv_heads"], world))
    if shape["ffn"] % (32 * world) or (shape["heads"] * shape["head_dim"]) % (32 * world):
        raise ValueError("ffn and heads*head_dim must split into whole 32-element blocks per rank")
    if shape["vocab"] % world:
        raise ValueError("vocab must be divisible by the group size")


def shard_shape(shape, world):
    s = dict(shape)
    s["heads"] = shape["heads"] // world
    s["kv_heads"] = shape["kv_heads"] // world
    s["ffn"] = shape["ffn"] // world
    return s


def slice_tensor(tid, full, rank, world):
    """full: 2-D tensor/array [rows][cols] of one weight; returns this rank's slice (a view)."""
    rows, cols = full.shape
    if tid in ROW_SPLIT:
        n = rows // world
        return full[rank * n:(rank + 1) * n]
    if tid in COL_SPLIT:
        n = cols // world
        return full[:, rank * n:(rank + 1) * n]
    return full


def build_tp_worker(shape_name, wdtype, kv_dtype, max_ctx, world, rank, device=0, std=0.02, **overrides):
    full = dict(synth.SHAPES[shape_name])
    full.update({k: v for k, v in overrides.items() if k in full})
    check_divisible(full, world)
    s = shard_shape(full, world)
    wk = W.DecodeWorker(max_ctx=max_ctx, kv_dtype=kv_dtype, device=device, tp_rank=rank, tp_size=world, **s)
    dev = "cuda:%d" % device

    def put(layer, tid, target, t16):
        t16 = t16.contiguous()
        rows, cols = (1, t16.numel()) if t16.dim() == 1 else t16.shape
        wk.set_tensor_f16(layer, tid, target, t16, rows, cols)

    put(-1, W.T_EMBD, dt.F16, synth.gen_f16((full["vocab"], full["dim"]), 999, std, dev))
    put(-1, W.T_OUT_NORM, dt.F16, torch.ones(full["dim"], dtype=torch.float16, device=dev))
    lm = synth.gen_f16((full["vocab"], full["dim"]), 998, std, dev)
    vs = full["vocab"] // world
    put(-1, W.T_LM_HEAD, dt.F16, lm[rank * vs:(rank + 1) * vs])
    del lm
    for layer in range(full["layers"]):
        put(layer, W.T_ATTN_NORM, dt.F16, torch.ones(full["dim"], dtype=torch.float16, device=dev))
        put(layer, W.T_FFN_NORM, dt.F16, torch.ones(full["dim"], dtype=torch.float16, device=dev))
        for tid, kind in synth.MATRICES:
            rows, cols = synth._shape(kind, full)
            t16 = synth.gen_f16((rows, cols), 1000 + layer * 16 + tid, std, dev)   # same stream of values as N=1
            put(layer, tid, wdtype, slice_tensor(tid, t16, rank, world))
    wk.finalize()
    return wk, full, s


class TPRunner:
    """Greedy batch-1 decode over a tensor-parallel group (all ranks call the same methods)."""

    def __init__(self, shape_name, wdtype, kv_dtype, max_ctx, world, rank, local_rank, group=None, **overrides):
        import os
        self.world, self.rank, self.group = world, rank, group
        # IFA_FORCE_TP=1 with one rank: still issue the collectives (plumbing check on a 1-GPU box)
        self.force_collectives = bool(os.environ.get("IFA_FORCE_TP")) and dist.is_initialized()
        self.worker = None
        self.worker, self.shape, self.local_shape = build_tp_worker(shape_name, wdtype, kv_dtype, max_ctx, world, rank,
                                                                    device=local_rank, **overrides)
        ok, why = self.worker.fused_supported()
        if not ok:
            raise RuntimeError("fused decode path unavailable on rank %d: %s" % (rank, why))
        dev = "cuda:%d" % local_rank
        self.dev = dev
        # the worker enqueues on the same (side) stream as the collectives, so that they are ordered without
        # extra events and a whole step can be captured into one hipGraph
        self.stream = torch.cuda.Stream(device=dev)
        self.worker.set_stream(self.stream.cuda_stream)
        self.graph = None
        # capturable collectives exist on the nccl (= RCCL) backend only; gloo (CPU tests) stays eager
        self.use_graph = os.environ.get("IFA_TP_GRAPH", "1") != "0" and (
            not dist.is_initialized() or dist.get_backend(group) == "nccl")
        d, vs = self.shape["dim"], self.shape["vocab"] // world
        self.buf_a = torch.zeros(d, dtype=torch.float16, device=dev)
        self.buf_f = torch.zeros(d, dtype=torch.float16, device=dev)
        self.logits = torch.zeros(vs, dtype=torch.float16, device=dev)
        self.best = torch.zeros(2, dtype=torch.float32, device=dev)
        self.gathered = torch.zeros(world * 2, dtype=torch.float32, device=dev)
        self.tok_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.vs = vs
        self._rank_off = torch.tensor(float(rank * vs), dtype=torch.float32, device=dev)
        self._big = torch.tensor(3.0e9, dtype=torch.float32, device=dev)

    def _all_reduce(self, t):
        if self.world > 1 or self.force_collectives:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def step(self, token, pos):
        """One decode step; `token` < 0 / `pos` < 0 reuse the id / position held in the device state."""
        with torch.cuda.stream(self.stream):
            return self._step_body(token, pos)

    def _step_body(self, token, pos):
        wk = self.worker
        wk.tp_begin(token, pos)
        for l in range(self.shape["layers"]):
            wk.tp_attn(l, self.buf_a)
            self._all_reduce(self.buf_a)
            wk.tp_post_attn(l, self.buf_a)
            wk.tp_ffn(l, self.buf_f)
            self._all_reduce(self.buf_f)
            wk.tp_post_ffn(l, self.buf_f)
        wk.tp_logits(self.logits)
        # distributed greedy argmax: (max value, global index) per rank, first maximum wins
        v, i = torch.max(self.logits.float(), dim=0)
        self.best[0] = v
        self.best[1] = i.float() + self._rank_off
        if self.world > 1:
            parts = list(self.gathered.view(self.world, 2).unbind(0))
            dist.all_gather(parts, self.best, group=self.group)
            g = torch.stack(parts)
        else:
            g = self.best.view(1, 2)
        top = g[:, 0].max()
        cand = torch.where(g[:, 0] == top, g[:, 1], self._big)
        self.tok_dev[0] = cand.min().to(torch.int32)
        wk.tp_set_token(self.tok_dev)
        return self.tok_dev

    def prefill(self, prompt):
        """Feeds the prompt through the decode path one token at a time; returns the next token."""
        tok = None
        for i, t in enumerate(np.asarray(prompt, np.int32)):
            tok = self.step(int(t), i)
        return int(tok.item())

    def _capture(self):
        """One whole step (worker kernels + RCCL collectives + distributed argmax) as a hipGraph that reads
        and advances the token / position in device memory.  Any failure leaves the eager path in place."""
        import sys
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                tok = self._step_body(-1, -1)
                self.ring.index_copy_(0, self.ring_pos, tok)
                self.ring_pos.add_(1)
            torch.cuda.synchronize()
            self.graph = g
        except Exception as e:       # keep going eagerly: correctness does not depend on the graph
            print("inferflow_amd.tp: step capture unavailable (%r); running eager steps" % (e,), file=sys.stderr)
            self.graph = None
            self.use_graph = False
            torch.cuda.synchronize()

    def decode(self, tok, pos, n):
        if not hasattr(self, "ring") or self.ring.numel() < n:
            self.ring = torch.zeros(max(n, 1024), dtype=torch.int32, device=self.dev)
            self.ring_pos = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self.graph = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            self.ring_pos.zero_()
            # the state (token, position) is seeded by one eager step; the rest replays the captured step
            t = self._step_body(int(tok), pos)
            self.ring.index_copy_(0, self.ring_pos, t)
            self.ring_pos.add_(1)
        if n > 1 and self.use_graph and self.graph is None:
            # the capture itself executes nothing, but it runs after the eager step above has created every
            # communicator / workspace the collectives need
            self._capture()
        with torch.cuda.stream(self.stream):
            e0.record(self.stream)
            for i in range(1, n):
                if self.graph is not None:
                    self.graph.replay()
                else:
                    t = self._step_body(-1, pos + i)
                    self.ring.index_copy_(0, self.ring_pos, t)
                    self.ring_pos.add_(1)
            e1.record(self.stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) if n > 1 else 0.0
        return [int(x) for x in self.ring[:n].cpu().numpy()], ms

    def export_host_tensors(self):
        raise NotImplementedError("CPU baseline runs on the single-worker runner only")
