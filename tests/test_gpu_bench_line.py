"""-m gpu: bench.py's contract on one GPU -- ONE JSON line on stdout whatever RCCL prints, the multi-GPU fallback chain
(C path with the captured step -> C path with eager steps -> torch.distributed runner) with injected failures, and the
single-device loopback mode.  Small step counts: this checks the plumbing, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--prefill-lens", "", "--batch", "0",
                        "--steps", "8", "--warmup", "2", "--shape", "test_gqa"] + list(args), capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                   # exactly one line on stdout
    return json.loads(lines[0]), r.stderr


def test_single_gpu_line_has_the_contract_keys():
    j, _ = _run({})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 8 and j["value"] > 0 and j["config"]["fallbacks"] == []


@pytest.mark.parametrize("inject,expect", [("", "c-abi"), ("c-graph", "[eager steps]"), ("c-graph,c-eager", "torch.distributed")])
def test_forced_collectives_and_the_fallback_chain(inject, expect):
    """IFA_FORCE_TP=1 makes a one-rank job issue every collective of the multi-GPU step; IFA_BENCH_FAIL_MODES injects a failure
    into the named modes: the line must come from the next mode and list what failed."""
    env = {"IFA_FORCE_TP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}
    if inject:
        env["IFA_BENCH_FAIL_MODES"] = inject
    j, err = _run(env)
    assert expect in j["config"]["collectives"], j["config"]
    assert [f["mode"] for f in j["config"]["fallbacks"]] == [m for m in inject.split(",") if m]
    assert j["value"] > 0


def test_loopback_two_ranks_on_one_device():
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--loopback", "2", "--steps", "8", "--warmup", "2", "--shape", "test_gqa"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert "tp2 loopback" in j["config"]["parallelism"] and "partition_ranks = 2" in j["config"]["parallelism"] and j["value"] > 0
