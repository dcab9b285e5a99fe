#!/bin/bash
# Round-3 profile set (run on the GPU box from the repo root; outputs under gpurun_out/prof_r03, copied to profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the bench                     -> r03_rocprofv3_kernel_stats.csv
#   2. PMC pass (own run, counters only): FETCH_SIZE                     -> r03_pmc_traffic.json (tools/pmc_summary.py)
#   3. the bench lines (after 2: they carry the traffic of THIS build)    -> r03_bench_n1.json, _q3h_q8, _q3h_f16, _mixtral
#   4. kernel stats of the persistent launch and of the batched step      -> r03_persist_kernel_stats.csv, r03_batch8_kernel_stats.csv
#   5. per-phase traces: persistent layer, rows GEMM of the batched step  -> r03_persist_phase_trace.log, r03_rows_gemm_phase_trace.log
#   6. dynamic batching 1..32 queries, N > 1 step logic on one device     -> r03_bench_batch.jsonl, r03_bench_loopback_tp2.json
set -x
OUT=$PWD/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python $R/bench.py --no-cpu-baseline --prefill-lens "" --batch 0 > $OUT/stats.log 2>&1)
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r03_rocprofv3_kernel_stats.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --prefill-lens "" --batch 0 > $OUT/pmc.log 2>&1)
python tools/pmc_summary.py $(find $OUT/pmc -name "*counter_collection.csv" | head -1) $OUT/r03_pmc_traffic.json > $OUT/pmc_summary.log 2>&1
cp $OUT/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 900 python bench.py > $OUT/r03_bench_n1.json 2> $OUT/bench.err
timeout 600 python bench.py --no-cpu-baseline --wdtype q3h --kv-dtype q8 > $OUT/r03_bench_n1_q3h_q8.json 2>> $OUT/bench.err
timeout 600 python bench.py --no-cpu-baseline --wdtype q3h > $OUT/r03_bench_n1_q3h_f16.json 2>> $OUT/bench.err
timeout 600 python bench.py --no-cpu-baseline --shape mixtral_8x7b --batch 8 --steps 64 > $OUT/r03_bench_mixtral.json 2>> $OUT/bench.err
timeout 600 python bench.py --loopback 2 --steps 32 --warmup 4 > $OUT/r03_bench_loopback_tp2.json 2>> $OUT/bench.err
IFA_BATCH_SIZES=1,2,4,8,16,17,24,32 timeout 600 python tools/bench_batch.py > $OUT/r03_bench_batch.jsonl 2>> $OUT/bench.err
(cd /tmp && IFA_BATCH_SIZES=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b8 -o b8 -- python $R/tools/bench_batch.py > $OUT/b8.log 2>&1)
cp $(find $OUT/b8 -name "*kernel_stats.csv" | head -1) $OUT/r03_batch8_kernel_stats.csv
timeout 300 python tools/trace_batch_rows.py 8 2>&1 | grep -A200 "=== step 2" | grep "rows-trace" | sed -n 49,64p > $OUT/r03_rows_gemm_phase_trace.log
timeout 300 python tools/debug_persist.py llama2_7b q4 f16 --steps 8 --trace 16 --time 64 > $OUT/r03_persist_phase_trace.log 2>&1
timeout 300 python tools/debug_persist.py llama2_7b q3h q8 --steps 8 --time 64 >> $OUT/r03_persist_phase_trace.log 2>&1
rm -rf $OUT/stats $OUT/pmc $OUT/b8
ls -la $OUT
tail -3 $OUT/bench.err
