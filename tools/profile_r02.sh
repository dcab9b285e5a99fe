#!/bin/bash
# Round-2 profile set (run on the GPU box from the repo root; outputs under gpurun_out/prof_r02, copied to profiles/ by hand):
#   1. rocprofv3 --kernel-trace --stats of the bench -> r02_rocprofv3_kernel_stats.csv
#   2. PMC pass (own run, counters only): FETCH_SIZE -> r02_pmc_traffic.json (tools/pmc_summary.py)
#   3. the bench line itself (after 2, so that it carries the traffic of THIS build) -> r02_bench_n1.json
#   4. PMC pass: SQ wave-cycle shares of the decode kernels -> r02_pmc_decode_sq_counters.txt (tools/sq_summary.py)
set -x
OUT=$PWD/gpurun_out/prof_r02
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python $OLDPWD/bench.py --no-cpu-baseline --prefill-lens "" > $OUT/stats.log 2>&1)
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r02_rocprofv3_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o pmc -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline --prefill-lens "" > $OUT/pmc.log 2>&1)
python tools/pmc_summary.py $(find $OUT/pmc -name "*counter_collection.csv" | head -1) $OUT/r02_pmc_traffic.json > $OUT/pmc_summary.log 2>&1
# the bench line AFTER the PMC pass: bench.py reports roofline.traffic from profiles/r02_pmc_traffic.json only when that file
# was taken with the kernel sources of the running build
cp $OUT/r02_pmc_traffic.json profiles/r02_pmc_traffic.json
python bench.py > $OUT/r02_bench_n1.json 2> $OUT/bench.err
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $OUT/sq -o sq -- python $OLDPWD/bench.py --steps 16 --warmup 2 --no-cpu-baseline --prefill-lens "" > $OUT/sq.log 2>&1)
python tools/sq_summary.py $(find $OUT/sq -name "*counter_collection.csv" | head -1) k_dec > $OUT/r02_pmc_decode_sq_counters.txt 2>&1
rm -rf $OUT/stats $OUT/pmc $OUT/sq
ls -la $OUT
