import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
from inferflow_amd import dtypes as dt, tp
PROMPT = np.array([5, 17, 400, 33, 2, 77], np.int32)

def main(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = tp.TPRunner("test_gqa", dt.Q4_B32T1A, dt.F16, 32, world, rank, 0, std=0.06, groups=2)
    for i, t in enumerate(PROMPT[:2]):
        tok = r.step(int(t), i)
        torch.cuda.synchronize()
        print("rank", rank, "step", i, "hidden", float(r.hidden.float().abs().sum()), "tok", int(tok.item()), "logits", float(r.logits.float().abs().sum()), flush=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    single = tp.TPRunner("test_gqa", dt.Q4_B32T1A, dt.F16, 32, 1, 0, 0, std=0.06)
    for i, t in enumerate(PROMPT[:2]):
        tok = single.step(int(t), i)
        torch.cuda.synchronize()
        print("single step", i, "tok", int(tok.item()), "logits", float(single.logits.float().abs().sum()), flush=True)
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=main, args=(r, 2, 29621)) for r in range(2)]
    [p.start() for p in ps]; [p.join(120) for p in ps]
