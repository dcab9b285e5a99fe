#!/bin/bash
# per-kernel rocprofv3 averages of a short prompt's prefill (GPU box, repo root): LENS="64 128" MINS="128 32"
OUT=$PWD/gpurun_out/prof_prompt
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for T in ${LENS:-64}; do for M in ${MINS:-128 32}; do
  (cd /tmp && IFA_PROMPT_LENS=$T IFA_BIG_MINS=$M timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o st -- python $R/tools/bench_prompt_lens.py > $OUT/p_${T}_$M.log 2>&1)
  f=$(find $OUT/p -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -30 $f > $OUT/prompt${T}_min${M}_kernel_stats.csv
  rm -rf $OUT/p
done; done
ls -la $OUT
