#!/bin/bash
# per-kernel rocprofv3 averages of the batched step at a few batch sizes and of the Mixtral batch-8 step (GPU box, repo root)
OUT=$PWD/gpurun_out/prof_batch
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for B in ${SIZES:-2 8 17 32}; do
  (cd /tmp && IFA_BATCH_SIZES=$B timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b$B -o st -- python $R/tools/bench_batch.py > $OUT/b$B.log 2>&1)
  f=$(find $OUT/b$B -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 $f > $OUT/batch${B}_kernel_stats.csv
  rm -rf $OUT/b$B
done
if [ -z "$NO_MIXTRAL" ]; then
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mx -o st -- python $R/bench.py --no-cpu-baseline --shape mixtral_8x7b --batch 8 --steps 4 --warmup 2 --prefill-lens "" > $OUT/mx.log 2>&1)
f=$(find $OUT/mx -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 $f > $OUT/mixtral_b8_kernel_stats.csv
rm -rf $OUT/mx
fi
ls -la $OUT
