"""CPU (gloo, world_size 2): the tensor-parallel partition rules of inferflow_amd.tp
reproduce the single-device result when the per-rank partial products -- computed
here with the ORACLE, since there is no GPU -- are summed across ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as o
    from inferflow_amd import dtypes as dt, tp, worker as W
    shape = dict(dim=256, layers=1, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=64)
    tp.check_divisible(shape, world)
    s = tp.shard_shape(shape, world)
    assert s["heads"] == 2 and s["kv_heads"] == 1 and s["ffn"] == 256
    rng = np.random.default_rng(5)                      # same full tensors on every rank
    x = rng.normal(0, 1.0, (1, 256)).astype(np.float16)
    w1 = rng.normal(0, 0.06, (512, 256)).astype(np.float16)
    w2 = rng.normal(0, 0.06, (256, 512)).astype(np.float16)
    d = dt.Q4_B32T1A
    # column-parallel then row-parallel pair (w1 rows split, w2 columns split)
    w1_r = np.ascontiguousarray(tp.slice_tensor(W.T_W1, w1, rank, world))
    w2_r = np.ascontiguousarray(tp.slice_tensor(W.T_W2, w2, rank, world))
    assert w1_r.shape == (256, 256) and w2_r.shape == (256, 256)
    xq = o.quantize_act_q8(x)
    t_r = o.gemv_ax8(d, o.quantize(d, w1_r), 256, 256, xq)            # this rank's 256 FFN rows
    part = o.gemv_ax8(d, o.quantize(d, w2_r), 256, 256, o.quantize_act_q8(t_r[None, :]))
    red = torch.from_numpy(part.astype(np.float32))
    dist.all_reduce(red)                                              # the exchange step
    # single-device reference
    t_full = o.gemv_ax8(d, o.quantize(d, w1), 512, 256, xq)
    full = o.gemv_ax8(d, o.quantize(d, w2), 256, 512, o.quantize_act_q8(t_full[None, :]))
    # slicing at block boundaries keeps every quant block identical, so the only
    # difference is one extra fp16 rounding of each partial sum
    err = np.abs(red.numpy() - full.astype(np.float32)).max()
    gathered = [None] * world
    dist.all_gather_object(gathered, t_r.tobytes())
    t_cat = np.concatenate([np.frombuffer(b, np.float16) for b in gathered])
    q.put((rank, float(err), bool(np.array_equal(t_cat.view(np.uint16), t_full.view(np.uint16)))))
    dist.destroy_process_group()


def test_tp_partition_matches_single_device():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "rank process failed (exit code %r)" % (p.exitcode,)
    res = [q.get(timeout=10) for _ in procs]
    for rank, err, rows_equal in res:
        assert rows_equal, "row-split GEMV outputs must be bit-identical to the unsplit ones"
        assert err <= 4e-3, "column-split partial sums differ from the unsplit product by %g" % err


def test_divisibility_rules():
    sys.path.insert(0, ROOT)
    from inferflow_amd import tp, synth
    tp.check_divisible(synth.SHAPES["llama2_7b"], 8)
    with pytest.raises(ValueError):
        tp.check_divisible(dict(synth.SHAPES["llama2_7b"], kv_heads=4), 8)
    s = tp.shard_shape(synth.SHAPES["llama2_7b"], 8)
    assert (s["heads"], s["kv_heads"], s["ffn"]) == (4, 4, 1376)


def test_split_layers_follows_the_reference_rule():
    # NetworkBuilder::SplitGpuLayers (network_builder.cc:2094-2118): ceil(L/G) per group, last group takes the rest,
    # empty groups dropped
    from inferflow_amd import tp
    assert tp.split_layers(32, 1) == [(0, 32)]
    assert tp.split_layers(32, 2) == [(0, 16), (16, 32)]
    assert tp.split_layers(60, 8) == [(0, 8), (8, 16), (16, 24), (24, 32), (32, 40), (40, 48), (48, 56), (56, 60)]
    assert tp.split_layers(6, 4) == [(0, 2), (2, 4), (4, 6)]        # 4th group would be empty
    assert tp.split_layers(5, 2) == [(0, 3), (3, 5)]
    for L in range(1, 40):
        for G in range(1, 9):
            r = tp.split_layers(L, G)
            assert r[0][0] == 0 and r[-1][1] == L and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def test_engine_partition_rules_match_the_python_partition():
    """The C++ engine's multi-GPU partition (host/model_loader.cc: SliceForWorker, SplitGpuLayers -- what `devices =
    0&1;2&3` loads onto each worker) against inferflow_amd.tp (slice_tensor / split_layers), which the gloo test above and
    the GPU tests exercise; host-only entry points of the C ABI."""
    import ctypes as C
    import inferflow_amd as ia
    from inferflow_amd import tp, worker as W
    L = ia.lib()
    for n_layers, groups in [(32, 1), (32, 2), (60, 4), (7, 2), (5, 3), (2, 4)]:
        out = (C.c_int * 16)()
        n = L.ifa_partition_split_layers(n_layers, groups, out, 8)
        got = [(out[2 * i], out[2 * i + 1]) for i in range(n)]
        assert got == tp.split_layers(n_layers, groups), (n_layers, groups)
    rows, cols = 64, 96
    full = np.arange(rows * cols, dtype=np.int64).reshape(rows, cols)
    out5 = (C.c_size_t * 5)()
    for P in (1, 2, 4):
        for r in range(P):
            for tid in (W.T_WQ, W.T_WK, W.T_WV, W.T_W1, W.T_W3, W.T_WO, W.T_W2, W.T_ATTN_NORM):
                assert L.ifa_partition_slice(0, 1, r, P, 0, 4, 2, tid, rows, cols, out5) == 1
                exp = tp.slice_tensor(tid, full, r, P)
                got = full[out5[0]:out5[1], out5[2]:out5[3]]
                assert np.array_equal(got, exp) and out5[4] == 2, (P, r, tid)
            # lm_head: vocabulary rows; biases of row-split matrices: the matching element range of the [1][n] vector
            assert L.ifa_partition_slice(0, 1, r, P, 0, 4, -1, W.T_LM_HEAD, rows, cols, out5) == 1
            assert (out5[0], out5[1], out5[2], out5[3]) == (r * rows // P, (r + 1) * rows // P, 0, cols)
            assert L.ifa_partition_slice(0, 1, r, P, 0, 4, 1, W.T_W1_B, 1, cols, out5) == 1
            assert (out5[0], out5[1], out5[2], out5[3]) == (0, 1, r * cols // P, (r + 1) * cols // P)
            assert L.ifa_partition_slice(0, 1, r, P, 0, 4, 1, W.T_WO_B, 1, cols, out5) == 1      # added once after the merge: replicated
            assert (out5[2], out5[3]) == (0, cols)
    # layer groups: a worker holds its range only, embeddings on the first group, lm_head / output norm on the last
    assert L.ifa_partition_slice(1, 2, 0, 1, 16, 32, 3, W.T_WQ, rows, cols, out5) == 0
    assert L.ifa_partition_slice(1, 2, 0, 1, 16, 32, 20, W.T_WQ, rows, cols, out5) == 1 and out5[4] == 4
    assert L.ifa_partition_slice(1, 2, 0, 1, 16, 32, -1, W.T_EMBD, rows, cols, out5) == 0
    assert L.ifa_partition_slice(0, 2, 0, 1, 0, 16, -1, W.T_EMBD, rows, cols, out5) == 1
    assert L.ifa_partition_slice(0, 2, 0, 1, 0, 16, -1, W.T_LM_HEAD, rows, cols, out5) == 0
    assert L.ifa_partition_slice(1, 2, 0, 1, 16, 32, -1, W.T_LM_HEAD, rows, cols, out5) == 1
