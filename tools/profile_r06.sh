#!/bin/bash
# Round-6 profile set (GPU box, repo root; outputs under gpurun_out/prof_r06, the summaries are copied to profiles/ afterwards).
# Run LAST: the PMC summary carries the hash of csrc/ (bench.py reports traffic: null for any other build).
#   1. rocprofv3 --kernel-trace --stats of the bench (the driver's command: --steps 20 --warmup 5)  -> r06_rocprofv3_kernel_stats.csv
#   2. PMC passes (own runs, counters only): FETCH_SIZE, headline config and configs[2]              -> r06_pmc_traffic.json, r06_pmc_q3h_q8_traffic.json
#   3. the bench lines (after 2: they carry the traffic of THIS build)                                -> r06_bench_n1.json (+ _steps128, _q3h_q8, _q3h_f16)
#   4. per-phase traces of the fused launch and the four GEMVs                                        -> r06_fused_launch_phase_trace.log, r06_kernel_phase_trace.log
#   5. A / B of the options that are still on the default path                                        -> r06_ab_options.log
#   6. dynamic batching 1..32 queries, Mixtral batch 8                                                -> r06_bench_batch.jsonl, r06_bench_mixtral.json
#   7. rocprofv3 averages of four 1024-token prefills                                                 -> r06_prefill_1024_kernel_stats.csv
#   8. prefill by prompt length, default routes                                                        -> r06_prefill_by_prompt_length.log
#   9. rocprofv3 averages of six 128-token prefills (the mid-size GEMM route)                         -> r06_prefill_128_kernel_stats.csv
set -x
OUT=$PWD/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --prefill-lens "" --batch 0 > $OUT/stats.log 2>&1)
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r06_rocprofv3_kernel_stats.csv
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
timeout 300 python tools/trace_fused.py > $OUT/r06_fused_launch_phase_trace.log 2>&1
timeout 300 python tools/trace_kernels.py > $OUT/r06_kernel_phase_trace.log 2>&1
(for o in fuse_attn attn_kt step_tail; do timeout 300 python tools/ab_option.py $o --steps 20 --prompt 21 --kernels; done) > $OUT/r06_ab_options.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --shape mixtral_8x7b --batch 8 --steps 64 > $OUT/r06_bench_mixtral.json 2>> $OUT/bench.err
IFA_BATCH_SIZES=1,2,4,8,16,17,24,32 timeout 600 python tools/bench_batch.py > $OUT/r06_bench_batch.jsonl 2>> $OUT/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pfst -o st -- python $R/tools/prefill_steps.py llama2_7b 1024 4 > $OUT/pfst.log 2>&1)
head -24 $(find $OUT/pfst -name "*kernel_stats.csv" | head -1) > $OUT/r06_prefill_1024_kernel_stats.csv
IFA_AB_OPTION=prefill_mid IFA_BIG_MINS=1 IFA_PROMPT_LENS=16,32,33,40,48,64,96,128,192,256,320,512,1024 timeout 400 python tools/bench_prompt_lens.py 2>&1 | grep "^T=" > $OUT/r06_prefill_by_prompt_length.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf128 -o st -- python $R/tools/prefill_steps.py llama2_7b 128 6 > $OUT/pf128.log 2>&1)
head -16 $(find $OUT/pf128 -name "*kernel_stats.csv" | head -1) > $OUT/r06_prefill_128_kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc $OUT/pmcq $OUT/pfst $OUT/pf128
ls -la $OUT
tail -3 $OUT/bench.err
