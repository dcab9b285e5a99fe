set -x
OUT=$PWD/gpurun_out/prof_r06b
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --prefill-lens "" --batch 0 > $OUT/pmc.log 2>&1)
python tools/pmc_summary.py $(find $OUT/pmc -name "*counter_collection.csv" | head -1) $OUT/r06_pmc_traffic.json > $OUT/pmc_summary.log 2>&1
cp $OUT/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcq -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --prefill-lens "" --batch 0 --wdtype q3h --kv-dtype q8 > $OUT/pmcq.log 2>&1)
python tools/pmc_summary.py $(find $OUT/pmcq -name "*counter_collection.csv" | head -1) $OUT/r06_pmc_q3h_q8_traffic.json > $OUT/pmcq_summary.log 2>&1
cp $OUT/r06_pmc_q3h_q8_traffic.json profiles/r06_pmc_q3h_q8_traffic.json
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/r06_bench_n1.json 2> $OUT/bench.err
timeout 600 python bench.py --no-cpu-baseline > $OUT/r06_bench_n1_steps128.json 2>> $OUT/bench.err
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --wdtype q3h --kv-dtype q8 --prefill-lens "" --batch 0 > $OUT/r06_bench_n1_q3h_q8.json 2>> $OUT/bench.err
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --wdtype q3h --prefill-lens "" --batch 0 > $OUT/r06_bench_n1_q3h_f16.json 2>> $OUT/bench.err
rm -rf $OUT/pmc $OUT/pmcq
python -m pytest tests/test_gpu_fused_attn.py tests/test_gpu_bench_line.py -x -q 2>&1 | grep -E "passed|failed" > $OUT/tests.log
cat $OUT/tests.log
