for ta in 4 8; do echo "== TA=$ta"; IFA_MID_TA=$ta IFA_MID_TRACE=1 python - <<'PY' 2>&1 | grep k_gemm_mid | sed -n 5,8p
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from inferflow_amd import dtypes as dt, synth
wk, _, s = synth.build("llama2_7b", dt.Q4_B32T1A, dt.F16, max_ctx=1100, layers=2)
wk.set_option("prefill_mid_max", 4096)
toks = np.random.default_rng(1).integers(3, s["vocab"], 1024).astype(np.int32)
wk.forward(toks, 0)
PY
done
