"""-m gpu: the downgrade paths of ADVICE r4 / VERDICT r4 item 8 on the device.
  * the MO-copy allocation fails inside the step that asked for it (simulated: option debug_mo_alloc_fail): the SAME step must
    still answer -- a 20-query step (which only the MO kernels serve as a fused step) falls to the op-by-op rows;
  * model options rows_kparts = 0 / gemm_splitk = 0: the launches whose workgroups wait for partner workgroups are never picked,
    results agree with the default within the F16 rounding of the summation order."""
import numpy as np
import pytest
import torch

from inferflow_amd import dtypes as dt, synth
from tests import gpu_util as g

pytestmark = pytest.mark.gpu


def _prefill_slots(wk, V, n, seed):
    rng = np.random.default_rng(seed)
    cur, pos = [], []
    for i in range(n):
        pr = rng.integers(3, V, 3 + (i * 5) % 9).astype(np.int32)
        wk.select_kv(i)
        cur.append(wk.forward(pr, 0)); pos.append(len(pr))
    return cur, pos


def _close(a, b):
    a, b = a.astype(np.float32), b.astype(np.float32)
    cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b)))
    return cos >= 0.9999 and np.abs(a - b).max() <= 0.02, (cos, float(np.abs(a - b).max()))


@pytest.mark.parametrize("n", [12, 20])
def test_failed_operand_order_copy_allocation_does_not_fail_the_step_that_asked_for_it(n):
    wk, _, s = synth.build("test_mha", dt.Q4_B32T1A, dt.F16, max_ctx=48, quant_threshold=0, std=0.06)
    ref, _, _ = synth.build("test_mha", dt.Q4_B32T1A, dt.F16, max_ctx=48, quant_threshold=0, std=0.06)
    V = s["vocab"]
    for w in (wk, ref):
        w.kv_slots(n)
    cur, pos = _prefill_slots(ref, V, n, 11)
    wk.set_option("rows_mo", 0)                      # (the prompts through the tiled kernels: nothing builds the copies yet)
    cur2, pos2 = _prefill_slots(wk, V, n, 11)
    assert cur2 == cur
    wk.set_option("rows_mo", 1)
    wk.set_option("debug_mo_alloc_fail", 1)          # the first batched step asks for the copies; the allocator "runs out"
    lg_a = torch.empty((n, V), dtype=torch.float16, device="cuda")
    lg_b = torch.empty((n, V), dtype=torch.float16, device="cuda")
    ta = wk.decode_batch(cur, pos, list(range(n)), lg_a)          # must answer (20 rows: op-by-op rows; 12: the tiled fused step)
    tb = ref.decode_batch(cur, pos, list(range(n)), lg_b)
    ok, why = _close(g.host(lg_a), g.host(lg_b))
    assert ok, why
    gaps = np.sort(g.host(lg_b).astype(np.float32), axis=1)
    for i in range(n):
        if gaps[i, -1] - gaps[i, -2] > 0.05:
            assert int(ta[i]) == int(tb[i]), i
    # and the steps after it (graph replay path included) keep answering
    cur = [int(t) for t in tb]; pos = [p + 1 for p in pos]
    tc = wk.decode_batch(cur, pos, list(range(n)))
    td = ref.decode_batch(cur, pos, list(range(n)))
    assert sum(int(a) == int(b) for a, b in zip(tc, td)) >= n - 2
    wk.close(); ref.close()


@pytest.mark.parametrize("n", [12, 24])
def test_rows_kparts_option_switches_the_waiting_launch_off(n):
    """w2 of test_longffn walks several chunks of K at 9..32 rows: K parts by default, plain chunk loop with rows_kparts = 0"""
    wk, _, s = synth.build("test_longffn", dt.Q4_B32T1A, dt.F16, max_ctx=48, quant_threshold=0, std=0.06)
    V = s["vocab"]
    wk.kv_slots(n)
    cur, pos = _prefill_slots(wk, V, n, 23)
    out = {}
    for v in (1, 0):
        wk.set_option("rows_kparts", v)
        lg = torch.empty((n, V), dtype=torch.float16, device="cuda")
        out[v] = (wk.decode_batch(cur, pos, list(range(n)), lg), g.host(lg).copy())      # (same position: the step rewrites the same cache rows)
    ok, why = _close(out[1][1], out[0][1])
    assert ok, why
    wk.close()


@pytest.mark.parametrize("n", [17, 24])
def test_kparts_launches_inside_a_captured_batched_step(n):
    """The K-parts launches of a 17..32-query step under stream capture (no logits requested: the step is captured and replayed) --
    found by tools/bench_batch.py in round 5: the launcher allocated its wait-error word (pinned host memory) at first use, which
    inside a thread-local capture fails and poisons the capture.  Captured and replayed steps must equal the eager step's ids."""
    wk, _, s = synth.build("test_longffn", dt.Q4_B32T1A, dt.F16, max_ctx=48, quant_threshold=0, std=0.06)
    V = s["vocab"]
    wk.kv_slots(n)
    cur, pos = _prefill_slots(wk, V, n, 29)
    lg = torch.empty((n, V), dtype=torch.float16, device="cuda")
    eager = wk.decode_batch(cur, pos, list(range(n)), lg)                 # eager (logits requested)
    first = wk.decode_batch(cur, pos, list(range(n)))                     # captures the step, runs it once
    again = wk.decode_batch(cur, pos, list(range(n)))                     # replay
    assert [int(t) for t in first] == [int(t) for t in eager] == [int(t) for t in again]
    wk.close()


def test_gemm_splitk_option_switches_the_waiting_launch_off():
    """a 256-token prompt at dim 4096-class widths takes 128 x 128 tiles in two halves of K by default; gemm_splitk = 0: whole K"""
    wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=320, layers=1, vocab=2000)
    V = s["vocab"]
    prompt = np.random.default_rng(3).integers(3, V, 256).astype(np.int32)
    out = {}
    for v in (1, 0):
        wk.set_option("gemm_splitk", v)
        wk.reset()
        lg = torch.empty((256, V), dtype=torch.float16, device="cuda")
        out[v] = (wk.forward(prompt, 0, lg), g.host(lg)[-1].copy())
    ok, why = _close(out[1][1], out[0][1])
    assert ok, why
    wk.close()
