"""-m gpu: the one-shot all-reduce between PROCESSES (what `torchrun` ranks are): every rank exports the IPC handles of its inbox
and flags (ifa_comm_oneshot_export), imports the peers' (hipIpcOpenMemHandle) and runs 30 all-reduces of decode-size vectors;
results must equal the rank-order half sums bit for bit.  The ranks share device 0 -- cross-process mapping, epoch / parity
protocol and arithmetic are what this covers; visibility between two DEVICES needs a multi-GPU box (bench.py compares the path
against RCCL there before using it)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("n", [2, 3])
def test_one_shot_all_reduce_between_processes(tmp_path, n):
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "oneshot_ipc_worker.py"), str(tmp_path), str(r), str(n)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(n)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for r, (rc, out) in enumerate(outs):
        assert rc == 0 and ("rank %d ok" % r) in out, out[-1500:]
