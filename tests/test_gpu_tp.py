"""-m gpu: tensor-parallel decode segments.  world=1 must reproduce the fused
single-worker decode bit for bit; world=2 (two processes sharing cuda:0, gloo
for the exchange because RCCL refuses two ranks on one device) must agree with
the single-device logits within the partial-sum rounding tolerance."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from inferflow_amd import dtypes as dt, synth, tp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROMPT = np.array([5, 17, 400, 33, 2, 77], np.int32)


def test_tp_world1_equals_fused_decode():
    wk, _, s = synth.build("test_gqa", dt.Q4_B32T1A, dt.F16, max_ctx=48, quant_threshold=0, std=0.06)
    tok = wk.forward(PROMPT, 0)
    ref, _ = wk.decode(tok, len(PROMPT), 10)
    wk.close()
    r = tp.TPRunner("test_gqa", dt.Q4_B32T1A, dt.F16, 48, 1, 0, 0, std=0.06)
    tok2 = r.prefill(PROMPT)
    got, ms = r.decode(tok2, len(PROMPT), 10)
    # prefill differs (op-by-op T>1 vs decode-path feed) only by fp tolerance; decode from the
    # same token must then be identical if the first tokens agree
    if tok2 == tok:
        assert got == [int(t) for t in ref]
    assert ms > 0


def _rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = tp.TPRunner("test_gqa", dt.Q4_B32T1A, dt.F16, 32, world, rank, 0, std=0.06)
    for i, t in enumerate(PROMPT):
        r.step(int(t), i)
    torch.cuda.synchronize()
    shards = [torch.zeros_like(r.logits) for _ in range(world)]
    dist.all_gather(shards, r.logits)
    if rank == 0:
        q.put(torch.cat(shards).float().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.close(); q.join_thread()


def test_tp_world2_matches_single_device_logits():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0, "rank process failed (exit code %r)" % (p.exitcode,)
    lg_tp = q.get(timeout=10)
    single = tp.TPRunner("test_gqa", dt.Q4_B32T1A, dt.F16, 32, 1, 0, 0, std=0.06)
    for i, t in enumerate(PROMPT):
        single.step(int(t), i)
    torch.cuda.synchronize()
    lg_1 = single.logits.float().cpu().numpy()
    cos = float((lg_tp * lg_1).sum() / (np.linalg.norm(lg_tp) * np.linalg.norm(lg_1)))
    # extra fp16 rounding of the two partial sums per layer, re-quantised downstream
    tol = 0.02 * float(np.abs(lg_1).max()) + 0.02     # ~2 % of the logit range (|logit| up to ~4 here)
    assert cos >= 0.9995 and np.abs(lg_tp - lg_1).max() <= tol, (cos, np.abs(lg_tp - lg_1).max(), tol)
