#!/bin/bash
# per-kernel rocprofv3 averages of a short prompt's prefill: LENS="128 256"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in ${LENS:-128}; do
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o st -- python $R/tools/prefill_steps.py llama2_7b $T 6 > /tmp/pp_$T.log 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
echo "== T=$T"; grep prefill /tmp/pp_$T.log | tail -2; grep -v "at::native\|rocclr\|k_quantize\|k_repack" $f | head -14 | cut -c1-200
cp $f $R/gpurun_out/r06_prefill_${T}_kernel_stats${TAG}.csv
done
