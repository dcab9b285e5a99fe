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

HYBRID / BY_LAYER (MultiGpuStrategy, src/transformer/model.h:61-66): `groups` device groups each own a
contiguous layer range (NetworkBuilder::SplitGpuLayers, network_builder.cc:2094-2118: ceil(L/G) layers per
group, the last group takes the rest); inside a group the layers are tensor-parallel as above.  rank =
group * group_size + tp_rank, the reference's "devices = 0&1;2&3" order.  The [dim] F16 layer output goes
to the same tp_rank of the next group with a send/recv pair (the reference's DeviceCopy between workers,
inference_worker.cc:2300-2335); the last group owns the lm_head and broadcasts the chosen token.
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


def split_layers(n_layers, groups):
    """[(start, end)] per device group, NetworkBuilder::SplitGpuLayers (network_builder.cc:2094-2118);
    groups that would be empty are dropped like the reference does."""
    per = (n_layers + groups - 1) // groups
    out = []
    for g in range(groups):
        start = g * per
        end = n_layers if g + 1 == groups else min((g + 1) * per, n_layers)   # (the reference does not clamp: it
        if end > start:                                                        #  assumes groups divide the layers)
            out.append((start, end))
    return out


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


def build_tp_worker(shape_name, wdtype, kv_dtype, max_ctx, world, rank, device=0, std=0.02, layer_range=None,
                    first_stage=True, last_stage=True, **overrides):
    """world / rank: size of and position in the TENSOR-parallel group.  layer_range (start, end): the
    global layers this worker holds (BY_LAYER / HYBRID); the worker numbers them 0..n-1 locally."""
    full = dict(synth.SHAPES[shape_name])
    full.update({k: v for k, v in overrides.items() if k in full})
    check_divisible(full, world)
    s = shard_shape(full, world)
    l0, l1 = layer_range if layer_range is not None else (0, full["layers"])
    s["layers"] = l1 - l0
    wk = W.DecodeWorker(max_ctx=max_ctx, kv_dtype=kv_dtype, device=device, tp_rank=rank, tp_size=world, **s)
    dev = "cuda:%d" % device

    def put(layer, tid, target, t16, expert=-1):
        t16 = t16.contiguous()
        rows, cols = (1, t16.numel()) if t16.dim() == 1 else t16.shape
        wk.set_tensor_f16(layer, tid, target, t16, rows, cols, expert=expert)

    if first_stage:
        put(-1, W.T_EMBD, dt.F16, synth.gen_f16((full["vocab"], full["dim"]), 999, std, dev))
    if last_stage:
        put(-1, W.T_OUT_NORM, dt.F16, torch.ones(full["dim"], dtype=torch.float16, device=dev))
        lm = synth.gen_f16((full["vocab"], full["dim"]), 998, std, dev)
        vs = full["vocab"] // world
        put(-1, W.T_LM_HEAD, dt.F16, lm[rank * vs:(rank + 1) * vs])
        del lm
    for layer in range(l0, l1):
        put(layer - l0, W.T_ATTN_NORM, dt.F16, torch.ones(full["dim"], dtype=torch.float16, device=dev))
        put(layer - l0, W.T_FFN_NORM, dt.F16, torch.ones(full["dim"], dtype=torch.float16, device=dev))
        n_exp = full.get("experts", 0)
        for tid, kind in synth.MATRICES:
            if tid == W.T_W3 and not full.get("is_glu", 1):
                continue
            rows, cols = synth._shape(kind, full)
            if n_exp and tid in (W.T_W1, W.T_W2, W.T_W3):      # experts: sliced like the dense FFN, one set per expert
                for e in range(n_exp):
                    t16 = synth.gen_f16((rows, cols), 100000 + (layer * 64 + e) * 16 + tid, std, dev)
                    put(layer - l0, tid, wdtype, slice_tensor(tid, t16, rank, world), expert=e)
                continue
            t16 = synth.gen_f16((rows, cols), 1000 + layer * 16 + tid, std, dev)   # same stream of values as N=1
            put(layer - l0, tid, wdtype, slice_tensor(tid, t16, rank, world))
        if n_exp:       # replicated router
            put(layer - l0, W.T_MOE_GATE, dt.F16, synth.gen_f16((n_exp, full["dim"]), 1000 + layer * 16 + W.T_MOE_GATE, std, dev))
    wk.finalize()
    return wk, full, s


class TPRunner:
    """Greedy batch-1 decode over a tensor-parallel group (all ranks call the same methods)."""

    def __init__(self, shape_name, wdtype, kv_dtype, max_ctx, world, rank, local_rank, group=None, groups=1, **overrides):
        """world / rank: all processes of the job.  groups > 1: HYBRID partition, `groups` layer ranges of
        world // groups tensor-parallel ranks each (every rank must construct the runner: new_group is collective)."""
        import os
        if world % groups:
            raise ValueError("world size %d is not a multiple of the number of device groups %d" % (world, groups))
        n_layers = dict(synth.SHAPES[shape_name], **{k: v for k, v in overrides.items() if k == "layers"})["layers"]
        ranges = split_layers(n_layers, groups)
        groups = len(ranges)
        self.job_world, self.job_rank = world, rank
        self.n_stages, self.stage = groups, 0
        self.layer_range = (0, n_layers)
        if groups > 1:
            tp_size = world // groups
            self.stage, tp_rank = rank // tp_size, rank % tp_size
            my_group = None
            for g in range(groups):            # collective: every rank creates every group, in the same order
                pg = dist.new_group(list(range(g * tp_size, (g + 1) * tp_size))) if tp_size > 1 else None
                if g == self.stage:
                    my_group = pg
            self.layer_range = ranges[self.stage]
            self.prev_rank = rank - tp_size if self.stage > 0 else None
            self.next_rank = rank + tp_size if self.stage + 1 < groups else None
            self.token_src = (groups - 1) * tp_size          # first rank of the last group announces the token
            world, rank, group = tp_size, tp_rank, my_group
        self.world, self.rank, self.group = world, rank, group
        # IFA_FORCE_TP=1 with one rank: still issue the collectives (plumbing check on a 1-GPU box)
        self.force_collectives = bool(os.environ.get("IFA_FORCE_TP")) and dist.is_initialized()
        self.worker = None
        self.worker, self.shape, self.local_shape = build_tp_worker(
            shape_name, wdtype, kv_dtype, max_ctx, world, rank, device=local_rank, layer_range=self.layer_range,
            first_stage=self.stage == 0, last_stage=self.stage == self.n_stages - 1, **overrides)
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
        self.use_graph = os.environ.get("IFA_TP_GRAPH", "1") != "0" and self.n_stages == 1 and (
            not dist.is_initialized() or dist.get_backend(group) == "nccl")
        self.hidden = torch.zeros(self.shape["dim"], dtype=torch.float16, device=dev)   # stage-to-stage layer output
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

    # gloo (CPU / single-GPU tests) has no device send/recv: stage through the host there
    def _send(self, t, dst):
        if dist.get_backend() == "nccl":
            dist.send(t, dst=dst)
        else:
            dist.send(t.cpu(), dst=dst)

    def _recv(self, t, src):
        if dist.get_backend() == "nccl":
            dist.recv(t, src=src)
        else:
            h = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(h, src=src)
            t.copy_(h)

    def _all_reduce(self, t):
        if self.world > 1 or self.force_collectives:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def step(self, token, pos):
        """One decode step; `token` < 0 / `pos` < 0 reuse the id / position held in the device state."""
        with torch.cuda.stream(self.stream):
            return self._step_body(token, pos)

    def _step_body(self, token, pos):
        wk = self.worker
        if self.stage == 0:
            wk.tp_begin(token, pos)
        else:
            self._recv(self.hidden, self.prev_rank)
            wk.tp_begin_hidden(self.hidden, pos)
        for l in range(self.layer_range[1] - self.layer_range[0]):
            wk.tp_attn(l, self.buf_a)
            self._all_reduce(self.buf_a)
            wk.tp_post_attn(l, self.buf_a)
            wk.tp_ffn(l, self.buf_f)
            self._all_reduce(self.buf_f)
            wk.tp_post_ffn(l, self.buf_f)
        if self.n_stages > 1:
            if self.next_rank is not None:         # not the last group: hand the layer output on, then wait for the token
                wk.tp_hidden(self.hidden)
                self._send(self.hidden, self.next_rank)
                dist.broadcast(self.tok_dev, src=self.token_src)
                wk.tp_set_token(self.tok_dev)
                return self.tok_dev
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
        if self.n_stages > 1:
            dist.broadcast(self.tok_dev, src=self.token_src)
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
            # thread_local: the process group's watchdog thread keeps polling its events while this thread captures
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
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


class CTPRunner:
    """The same partition driven from C: every exchange is a collective of the C ABI (csrc/ifa_comm.hip, RCCL) enqueued on
    the worker's stream by ifa_model_tp_prefill / ifa_model_tp_decode, the step is one hipGraph.  torch.distributed only
    carries the 128-byte communicator ids at start-up (any host channel would do)."""

    backend = "c-abi (ifa_comm.hip: RCCL all-reduce / all-gather / send / recv / broadcast)"

    def __init__(self, shape_name, wdtype, kv_dtype, max_ctx, world, rank, local_rank, groups=1, **overrides):
        import os
        if world % groups:
            raise ValueError("world size %d is not a multiple of the number of device groups %d" % (world, groups))
        n_layers = dict(synth.SHAPES[shape_name], **{k: v for k, v in overrides.items() if k == "layers"})["layers"]
        ranges = split_layers(n_layers, groups)
        groups = len(ranges)
        tp_size = world // groups
        self.stage, self.tp_rank = rank // tp_size, rank % tp_size
        self.n_stages, self.world, self.rank = groups, world, rank
        self.force_collectives = bool(os.environ.get("IFA_FORCE_TP"))
        dev = "cuda:%d" % local_rank

        def share_id(src, make):
            """128-byte RCCL id from job rank `src` to everybody"""
            buf = torch.zeros(128, dtype=torch.uint8, device=dev)
            if make:
                buf.copy_(torch.frombuffer(bytearray(W.Comm.unique_id()), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(buf, src=src)
            return bytes(buf.cpu().numpy().tobytes())

        self.world_comm = None
        if groups > 1:
            self.world_comm = W.Comm(share_id(0, rank == 0), world, rank, local_rank)
        self.tp_comm = None
        if tp_size > 1 or self.force_collectives:
            ids = [share_id(g * tp_size, rank == g * tp_size) for g in range(groups)]     # collective: same order on every rank
            self.tp_comm = W.Comm(ids[self.stage], tp_size, self.tp_rank, local_rank)
            # IFA_ONESHOT_IPC=1: map the group's one-shot inboxes across the processes (csrc/ifa_comm.hip; the handles travel
            # like the communicator id) -- decode-size all-reduces then skip RCCL.  Opt-in: bench.py switches it on as its
            # first mode and compares a few steps against the RCCL path before timing with it.
            self.oneshot_ipc = False
            if os.environ.get("IFA_ONESHOT_IPC") == "1" and tp_size > 1 and world > 1:
                # every step below is taken by EVERY rank whatever happened to it locally (a rank that skipped a collective would
                # strand the others), and the outcome is agreed: one failure anywhere keeps RCCL everywhere
                ok, handle = 1, bytes(128)
                try:
                    handle = self.tp_comm.oneshot_export()
                except Exception:      # noqa: BLE001
                    ok = 0
                mine = torch.frombuffer(bytearray(handle), dtype=torch.uint8).to(dev)
                allh = [torch.zeros(128, dtype=torch.uint8, device=dev) for _ in range(world)]
                dist.all_gather(allh, mine)
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()):
                    base = self.stage * tp_size
                    try:
                        self.tp_comm.oneshot_import(b"".join(bytes(allh[base + r].cpu().numpy().tobytes()) for r in range(tp_size)))
                    except Exception:      # noqa: BLE001
                        ok = 0
                    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                self.oneshot_ipc = bool(int(flag.item())) and self.tp_comm.oneshot()
                if not self.oneshot_ipc:
                    self.tp_comm.set_oneshot(0)
        self.worker, self.shape, self.local_shape = build_tp_worker(
            shape_name, wdtype, kv_dtype, max_ctx, tp_size, self.tp_rank, device=local_rank, layer_range=ranges[self.stage],
            first_stage=self.stage == 0, last_stage=self.stage == groups - 1, **overrides)
        ok, why = self.worker.fused_supported()
        if not ok:
            raise RuntimeError("fused decode path unavailable on rank %d: %s" % (rank, why))
        self.topo = dict(tp=self.tp_comm, world=self.world_comm, stage=self.stage, n_stages=groups,
                         prev_rank=rank - tp_size if self.stage > 0 else -1,
                         next_rank=rank + tp_size if self.stage + 1 < groups else -1,
                         token_src=(groups - 1) * tp_size, vocab_offset=self.tp_rank * (self.shape["vocab"] // tp_size),
                         force_collectives=self.force_collectives)

    def prefill(self, prompt):
        import ctypes as C
        from ._capi import check, lib
        toks = np.ascontiguousarray(prompt, np.int32)
        topo = W.TpTopology(self.topo["tp"]._h if self.topo["tp"] else None, self.topo["world"]._h if self.topo["world"] else None,
                            self.topo["stage"], self.topo["n_stages"], self.topo["prev_rank"], self.topo["next_rank"],
                            self.topo["token_src"], self.topo["vocab_offset"], 1 if self.topo["force_collectives"] else 0)
        nxt = C.c_int(-1)
        check(lib().ifa_model_tp_prefill(self.worker._h, C.byref(topo), toks.ctypes.data_as(C.c_void_p), toks.size, 0, None, C.byref(nxt)))
        return nxt.value

    def decode(self, tok, pos, n):
        out, ms_total = [], 0.0
        while n > 0:
            k = min(n, 1024)
            t, ms = W.tp_decode(self.worker, tok, pos, k, **self.topo)
            out.extend(int(x) for x in t)
            ms_total += ms
            tok, pos, n = int(t[-1]), pos + k, n - k
        return out, ms_total

    def export_host_tensors(self):
        raise NotImplementedError("CPU baseline runs on the single-worker runner only")
