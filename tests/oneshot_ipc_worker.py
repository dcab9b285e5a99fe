"""One rank of the cross-process one-shot all-reduce test (tests/test_gpu_oneshot_ipc.py): python oneshot_ipc_worker.py <dir> <rank> <nranks>.
The ranks are separate PROCESSES sharing device 0 (RCCL refuses that, hence IFA_COMM_TEST_NO_RCCL: a communicator with the
one-shot side only); handles and barriers travel through files in <dir>."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IFA_COMM_TEST_NO_RCCL"] = "1"
import numpy as np
import torch

from inferflow_amd import worker as W

d, rank, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])


def barrier(tag, timeout=60.0):
    open(os.path.join(d, "%s_%d" % (tag, rank)), "w").write("x")
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, "%s_%d" % (tag, r))) for r in range(n)):
        if time.time() - t0 > timeout:
            raise RuntimeError("barrier %s timed out" % tag)
        time.sleep(0.005)


torch.cuda.set_device(0)
comm = W.Comm(b"\0" * 128, n, rank, 0)
h = comm.oneshot_export()
assert len(h) == 128 and not comm.oneshot()
open(os.path.join(d, "h_%d.tmp" % rank), "wb").write(h)
os.rename(os.path.join(d, "h_%d.tmp" % rank), os.path.join(d, "h_%d" % rank))
barrier("exported")
comm.oneshot_import(b"".join(open(os.path.join(d, "h_%d" % r), "rb").read() for r in range(n)))
assert comm.oneshot()
barrier("imported")


def vec(r, it, count):
    return (np.random.default_rng(1000 * it + r).normal(0, 1.0, count)).astype(np.float16)


for it, count in enumerate([4096, 8, 11008, 4096, 32768, 4100, 4096, 4096, 1, 5120] * 3):
    x = torch.from_numpy(vec(rank, it, count)).cuda()
    comm.all_reduce_f16(x)                         # in place: push to every peer's inbox, wait for the epoch, sum in rank order
    torch.cuda.synchronize()
    acc = vec(0, it, count)
    for r in range(1, n):                          # ((v0 + v1) + v2) ... in half: MergeTensors' order
        acc = (acc.astype(np.float32) + vec(r, it, count).astype(np.float32)).astype(np.float16)
    got = x.cpu().numpy()
    if not np.array_equal(got.view(np.uint16), acc.view(np.uint16)):
        raise SystemExit("rank %d: all-reduce %d (%d halves) differs in %d places" % (rank, it, count, int((got.view(np.uint16) != acc.view(np.uint16)).sum())))
assert comm.status() == 0
barrier("done")                                    # nobody frees its inbox while a peer may still write to it
comm.close()
print("rank %d ok" % rank)
