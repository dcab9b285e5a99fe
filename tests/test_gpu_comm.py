"""-m gpu: the collectives of the C ABI (csrc/ifa_comm.hip, RCCL) and the C-driven multi-GPU decode loop
(ifa_model_tp_decode).  A 1-GPU box can only form a communicator of ONE rank (RCCL refuses two ranks on a device), which
still runs every call through RCCL -- including inside the captured step; groups of 2+ ranks run where >= 2 GPUs are
visible and skip cleanly otherwise.  The partition arithmetic of larger groups is covered on CPU (tests/test_tp_cpu.py)
and with gloo ranks sharing one GPU (tests/test_gpu_tp.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from inferflow_amd import dtypes as dt, synth, tp
from inferflow_amd import worker as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROMPT = np.array([5, 17, 400, 33, 2, 77], np.int32)


def test_comm_of_one_rank_runs_every_collective():
    c = W.Comm(W.Comm.unique_id(), 1, 0, 0)
    assert c.nranks == 1
    x = torch.randn(4096, device="cuda").half()
    ref = x.clone()
    c.all_reduce_f16(x)
    g = torch.zeros(8, dtype=torch.uint8, device="cuda")
    src = torch.arange(8, dtype=torch.uint8, device="cuda")
    c.all_gather(src, g)
    c.broadcast(x, 0)
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and torch.equal(g, src)
    with pytest.raises(Exception):
        c.send(x, 0)          # a rank cannot send to itself: argument error, not a hang
    c.close()


def test_c_driven_tp_decode_equals_fused_decode_and_python_runner():
    """world = 1: the tensor-parallel shard IS the model, so the C-driven loop (segments + forced RCCL collectives + the
    distributed argmax, captured as one hipGraph per step) must give the fused single-worker decode's tokens bit for bit."""
    wk, _, s = synth.build("test_gqa", dt.Q4_B32T1A, dt.F16, max_ctx=48, quant_threshold=0, std=0.06)
    tok = wk.forward(PROMPT, 0)
    ref, _ = wk.decode(tok, len(PROMPT), 12)
    c = W.Comm(W.Comm.unique_id(), 1, 0, 0)
    wk.forward(PROMPT, 0)                                  # same KV prefix again
    got, ms = W.tp_decode(wk, tok, len(PROMPT), 12, tp=c, force_collectives=True)
    assert [int(t) for t in got] == [int(t) for t in ref] and ms > 0
    wk.forward(PROMPT, 0)
    got2, _ = W.tp_decode(wk, tok, len(PROMPT), 12)        # no communicator at all: plain segments
    assert [int(t) for t in got2] == [int(t) for t in ref]
    # excluded ids reach the distributed argmax too
    wk.set_excluded_tokens([int(ref[0])])
    wk.forward(PROMPT, 0)
    got3, _ = W.tp_decode(wk, tok, len(PROMPT), 4, tp=c, force_collectives=True)
    assert int(got3[0]) != int(ref[0])
    wk.set_excluded_tokens([])
    c.close(); wk.close()


def test_loopback_group_collectives_between_rank_threads():
    """ifa_comm_init_all with one device named n times: the in-process loopback group (what lets the multi-rank engine
    paths run on a 1-GPU box).  Rank threads call the collectives concurrently, like the engine's rank threads."""
    import ctypes as C
    import threading
    import inferflow_amd as ia
    L = ia.lib()
    n = 3
    devs = (C.c_int * n)(0, 0, 0)
    comms = (C.c_void_p * n)()
    ia.check(L.ifa_comm_init_all(devs, n, comms))
    assert L.ifa_comm_capturable(comms[0]) == 0 and L.ifa_comm_size(comms[1]) == n and L.ifa_comm_rank(comms[2]) == 2
    xs = [torch.full((1000,), float(r + 1), device="cuda").half() * 0.25 for r in range(n)]
    g_in = [torch.full((8,), r, dtype=torch.uint8, device="cuda") for r in range(n)]
    g_out = [torch.zeros(8 * n, dtype=torch.uint8, device="cuda") for _ in range(n)]
    h = [torch.full((64,), float(r), device="cuda").half() for r in range(n)]
    torch.cuda.synchronize()
    errs = []

    def rank(r):
        try:
            st = torch.cuda.Stream()
            sp = C.c_void_p(st.cuda_stream)
            ia.check(L.ifa_allreduce_sum_f16(comms[r], C.c_void_p(xs[r].data_ptr()), C.c_void_p(xs[r].data_ptr()), 1000, sp))
            ia.check(L.ifa_allgather(comms[r], C.c_void_p(g_in[r].data_ptr()), C.c_void_p(g_out[r].data_ptr()), 8, sp))
            ia.check(L.ifa_broadcast(comms[r], C.c_void_p(h[r].data_ptr()), 128, 1, sp))
            if r == 0:
                ia.check(L.ifa_send(comms[r], C.c_void_p(xs[r].data_ptr()), 2000, 2, sp))
            if r == 2:
                ia.check(L.ifa_recv(comms[r], C.c_void_p(xs[r].data_ptr()), 2000, 0, sp))
            st.synchronize()
        except Exception as e:      # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=rank, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert not errs and all(not t.is_alive() for t in ts)
    torch.cuda.synchronize()
    assert all(float(x[0]) == 1.5 for x in xs)                      # 0.25 + 0.5 + 0.75
    assert all(g.cpu().tolist() == [0] * 8 + [1] * 8 + [2] * 8 for g in g_out)
    assert all(float(t[0]) == 1.0 for t in h)
    for c in comms:
        L.ifa_comm_destroy(c)


@pytest.mark.parametrize("shape,tol", [("test_gqa", 0.03), ("test_moe", 0.05)], ids=["dense", "moe"])
def test_batched_decode_over_a_tensor_parallel_group_matches_oracle(shape, tol):
    """(moe: the Mixtral kind of step -- configs[4] -- experts sliced over the ranks like the dense FFN; the reference has no
    tensor-parallel MoE to restate: each rank accumulates its weighted expert products before the merge, the
    oracle merges per expert, one more half rounding apart: stated bound 0.05.)
    Dynamic batching under BY_TENSOR (configs[4] shape of work: several queries, one step, merges over [n][dim]): two
    rank threads on one GPU (loopback group), three queries on their own KV slots, every row against the oracle of its
    query with the merge restated and the T > 1 arithmetic (F16 activations) of a batched step."""
    import threading
    from inferflow_amd import tp as tpmod
    from tests.model_util import oracle_model_from_host
    P = 2
    wk1, host, s = synth.build(shape, dt.Q4_B32T1A, dt.F16, max_ctx=64, quant_threshold=0, std=0.06, keep_host=True)
    wk1.close()
    V = s["vocab"]
    workers = [tpmod.build_tp_worker(shape, dt.Q4_B32T1A, dt.F16, 64, P, r, device=0, std=0.06)[0] for r in range(P)]
    comms = W.Comm.init_all([0] * P)
    rng = np.random.default_rng(23)
    prompts = [rng.integers(3, V, n).astype(np.int32) for n in (5, 11, 8)]
    oms = [oracle_model_from_host(host, s, 64, dt.F16, full_quant_gemv=0, tp_merge=P) for _ in prompts]
    for wk in workers:
        wk.kv_slots(3)
    firsts = [[None] * 3 for _ in range(P)]
    shards = [torch.zeros((3, V // P), dtype=torch.float16, device="cuda") for _ in range(P)]
    outs = [None] * P
    errs = []

    def on_ranks(fn):
        ts = [threading.Thread(target=lambda r=r: _guard(fn, r, errs)) for r in range(P)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not errs, errs
        assert all(not t.is_alive() for t in ts)

    def prefill(r):
        for i, pr in enumerate(prompts):
            workers[r].select_kv(i)
            firsts[r][i] = W.tp_prefill(workers[r], pr, 0, tp=comms[r], vocab_offset=r * (V // P))
    on_ranks(prefill)
    assert firsts[0] == firsts[1]
    first_o = []
    for i, pr in enumerate(prompts):
        t_o, l_o = oms[i].forward(pr, 0, nthreads=4)
        first_o.append(t_o)
        top2 = np.sort(l_o[-1].astype(np.float32))[-2:]
        if top2[1] - top2[0] > tol:
            assert firsts[0][i] == t_o
    cur, pos = list(firsts[0]), [len(p) for p in prompts]
    for step in range(4):
        def one(r):
            outs[r] = W.tp_decode_batch(workers[r], cur, pos, [0, 1, 2], tp=comms[r], vocab_offset=r * (V // P), logits_shard_out=shards[r])
        on_ranks(one)
        assert list(outs[0]) == list(outs[1])
        rows = torch.cat(shards, 1).float().cpu().numpy()
        for i in range(3):
            t_o, l_o = oms[i].forward(np.array([cur[i]], np.int32), pos[i], nthreads=4)
            a, b = rows[i], l_o[0].astype(np.float32)
            cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b)))
            assert cos >= 0.9995 and np.abs(a - b).max() <= tol, (step, i, cos, np.abs(a - b).max())
            top2 = np.sort(b)[-2:]
            if top2[1] - top2[0] > tol:
                assert int(outs[0][i]) == t_o, (step, i)
        cur, pos = [int(t) for t in outs[0]], [p + 1 for p in pos]
    for c in comms:
        c.close()
    for wk in workers:
        wk.close()


def _guard(fn, r, errs):
    try:
        fn(r)
    except Exception as e:      # pragma: no cover
        errs.append((r, repr(e)))


def _rank(rank, world, uid, q):
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(rank)
    c = W.Comm(uid, world, rank, rank)
    x = torch.full((4096,), float(rank + 1), device="cuda:%d" % rank).half()
    c.all_reduce_f16(x)
    h = torch.full((256,), float(rank), device="cuda:%d" % rank).half()
    if rank == 0:
        c.send(h, 1)
    elif rank == 1:
        c.recv(h, 0)
    torch.cuda.synchronize()
    q.put((rank, float(x[0].item()), float(h[0].item())))
    c.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL refuses two ranks on one device)")
def test_allreduce_and_send_recv_over_two_gpus():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = W.Comm.unique_id()
    ps = [ctx.Process(target=_rank, args=(r, 2, uid, q)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert res[0][1] == 3.0 and res[1][1] == 3.0 and res[1][2] == 0.0


def _loopback_allreduce(n, count, oneshot, rounds=3):
    """n rank threads on ONE device, each with its own stream-less tensor: returns (results per round per rank, expected)"""
    import threading
    comms = W.Comm.init_all([0] * n)
    for c in comms:
        c.set_oneshot(oneshot)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234 + n + count)
    data = [[(torch.randn(count, generator=gen, device="cuda") * 3).half() for _ in range(n)] for _ in range(rounds)]
    out = [[None] * n for _ in range(rounds)]
    errs = []

    def run(r):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for k in range(rounds):
                    t = data[k][r].clone()
                    comms[r].all_reduce_f16(t, stream=s.cuda_stream)
                    s.synchronize()
                    out[k][r] = t
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    [t.start() for t in th]
    [t.join(60) for t in th]
    assert not errs, errs
    st = [c.status() for c in comms]
    used = comms[0].oneshot()
    for c in comms:
        c.close()
    exp = []
    for k in range(rounds):       # ((v0 + v1) + v2) ... in half: MergeTensors order (inference_worker.cc:2197-2260)
        acc = data[k][0].clone()
        for r in range(1, n):
            acc = (acc.float() + data[k][r].float()).half()
        exp.append(acc)
    return out, exp, st, used


@pytest.mark.parametrize("n,count", [(2, 4096), (4, 4096), (4, 8192), (3, 1001)])
def test_oneshot_allreduce_sums_in_rank_order_like_the_rendezvous_path(n, count):
    """The peer-mapped one-shot all-reduce (ranks sharing this box's one GPU map each other's memory trivially): every
    rank gets the rank-order half sum, bit for bit, over several epochs (both inbox halves); the host-rendezvous path of
    the same group gives the same bits.  (At most 4 ranks here: ranks sharing a device need a hardware queue each to run at
    the same time, which is why the exchange is opt-in for loopback groups.)"""
    out, exp, st, used = _loopback_allreduce(n, count, True)
    assert used, "one-shot exchange was not set up"
    assert st == [0] * n
    for k in range(len(exp)):
        for r in range(n):
            assert torch.equal(out[k][r], exp[k]), (k, r)
    out2, exp2, _, used2 = _loopback_allreduce(n, count, False)
    assert not used2
    for k in range(len(exp2)):
        for r in range(n):
            assert torch.equal(out2[k][r], exp2[k]), (k, r)


def test_abort_wakes_a_rank_blocked_in_a_collective():
    """A peer that fails never reaches the collective: ifa_comm_abort makes the waiting rank return an error instead of
    blocking forever (the host engine calls it from the failing rank's thread, host/inference_engine.cc)."""
    import threading
    import time
    comms = W.Comm.init_all([0, 0])
    for c in comms:
        c.set_oneshot(False)
    res = {}

    def waiter():
        torch.cuda.set_device(0)
        x = torch.ones(64, device="cuda").half()
        try:
            comms[0].all_reduce_f16(x)
            res["ok"] = True
        except Exception as e:  # noqa: BLE001
            res["err"] = repr(e)

    t = threading.Thread(target=waiter)
    t.start()
    time.sleep(0.3)
    assert t.is_alive()                 # blocked in the rendezvous: rank 1 never comes
    comms[1].abort()
    t.join(10)
    assert not t.is_alive() and "err" in res and "abort" in res["err"]
    for c in comms:
        c.close()


def test_moe_decode_over_a_tensor_parallel_group_follows_the_oracle():
    """MoE + tensor parallelism at T = 1 (the fused MoE decode kernels under the merges): two rank threads on one GPU,
    a prompt through the partition, then 8 greedy steps of ifa_model_tp_decode -- next-token ids against the oracle (merge
    restated) wherever its top-2 gap exceeds the stated 0.05, and the ranks agree step for step."""
    import threading
    from inferflow_amd import tp as tpmod
    from tests.model_util import oracle_model_from_host
    P = 2
    wk1, host, s = synth.build("test_moe", dt.Q4_B32T1A, dt.F16, max_ctx=64, quant_threshold=0, std=0.06, keep_host=True)
    wk1.close()
    V = s["vocab"]
    workers = [tpmod.build_tp_worker("test_moe", dt.Q4_B32T1A, dt.F16, 64, P, r, device=0, std=0.06)[0] for r in range(P)]
    comms = W.Comm.init_all([0] * P)
    om = oracle_model_from_host(host, s, 64, dt.F16, full_quant_gemv=1, tp_merge=P)
    prompt = np.random.default_rng(31).integers(3, V, 9).astype(np.int32)
    first, toks, errs = [None] * P, [None] * P, []

    def run(r):
        first[r] = W.tp_prefill(workers[r], prompt, 0, tp=comms[r], vocab_offset=r * (V // P))
        toks[r], _ = W.tp_decode(workers[r], first[r], len(prompt), 8, tp=comms[r], vocab_offset=r * (V // P))

    ts = [threading.Thread(target=lambda r=r: _guard(run, r, errs)) for r in range(P)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert not errs, errs
    assert first[0] == first[1] and [int(t) for t in toks[0]] == [int(t) for t in toks[1]]
    t_o, l_o = om.forward(prompt, 0, nthreads=4)
    cur, pos, checked = int(first[0]), len(prompt), 0
    for i in range(8):
        t_o, l_o = om.forward(np.array([cur], np.int32), pos, nthreads=4)
        top2 = np.sort(l_o[0].astype(np.float32))[-2:]
        if top2[1] - top2[0] > 0.05:
            assert int(toks[0][i]) == t_o, i
            checked += 1
        cur, pos = int(toks[0][i]), pos + 1
    assert checked >= 4
    for c in comms:
        c.close()
    for wk in workers:
        wk.close()
