import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from inferflow_amd import dtypes as dt, synth
from tests import gpu_util as g
for T in (2, 8):
    wk, host, s = synth.build("test_moe", dt.Q4_B32T1A, dt.F16, max_ctx=64, quant_threshold=0, std=0.06, keep_host=True)
    prompt = np.random.default_rng(100 + T).integers(3, s["vocab"], T).astype(np.int32)
    lg = torch.empty((T, s["vocab"]), dtype=torch.float16, device="cuda")
    res = {}
    for name, md, gr in (("host", 0, 1), ("dev_rows", 1, 1), ("dev_norows", 1, 0)):
        wk.set_option("moe_device", md); wk.set_option("gemm_rows", gr)
        wk.forward(prompt, 0, lg)
        res[name] = g.host(lg).astype(np.float32).copy()
    for k in ("dev_rows", "dev_norows"):
        print(T, k, np.abs(res[k] - res["host"]).max())
    wk.close()
