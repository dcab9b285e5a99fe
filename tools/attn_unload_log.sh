#!/bin/bash
# the measurements behind profiles/r06_attn_unload.log (GPU box, repo root)
OUT=gpurun_out/attn_unload; mkdir -p $OUT; L=$OUT/log; rm -f $L
echo "# per-launch HIP-event averages (repeated launches: cache rows WARM) next to the rate of a real 16-step call (rows COLD), by context: tools/ctx_kernels.py (attn_unload 0)" >> $L
IFA_CTXK_UNLOAD=0 timeout 300 python tools/ctx_kernels.py 2>&1 | grep context >> $L
echo "# option attn_unload 0 / 1, one process, alternating: tools/attn_unload_ab.py (F16 cache, Q8 cache, Mixtral-8x7B)" >> $L
timeout 300 python tools/attn_unload_ab.py 2>&1 | grep context >> $L
timeout 300 python tools/attn_unload_ab.py q8 2>&1 | grep context >> $L
timeout 500 python tools/attn_unload_ab.py mixtral_8x7b 2>&1 | grep context >> $L
echo "# the 128-row bucket too (attn_unload 2) against 1: slower below ~110 keys" >> $L
IFA_AB_CTX=60,80,100,110 IFA_AB_UL=1,2 timeout 300 python tools/attn_unload_ab.py 2>&1 | grep context >> $L
echo "# one workgroup per head against keys split over workgroups (attn_split_ctx 320 / 1000): tools/split_threshold_ab.py" >> $L
timeout 300 python tools/split_threshold_ab.py 2>&1 | grep context >> $L
timeout 300 python tools/split_threshold_ab.py q8 2>&1 | grep context >> $L
timeout 500 python tools/split_threshold_ab.py mixtral_8x7b 2>&1 | grep context >> $L
echo "# decode by context, default options: tools/ctx_sweep.py" >> $L
timeout 300 python tools/ctx_sweep.py 2>&1 | grep context >> $L
timeout 300 python tools/ctx_sweep.py q8 2>&1 | grep context >> $L
cat $L | tail -30
