#!/bin/bash
# Round-4 profile set (run on the GPU box from the repo root; outputs under gpurun_out/prof_r04, copied to profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the bench (the driver's command: --steps 20 --warmup 5) -> r04_rocprofv3_kernel_stats.csv
#   2. PMC pass (own run, counters only): FETCH_SIZE                                             -> r04_pmc_traffic.json (tools/pmc_summary.py)
#   3. the bench lines (after 2: they carry the traffic of THIS build)                            -> r04_bench_n1.json (+ _steps128, _q3h_q8, _q3h_f16)
#   4. per-phase traces of the fused launches and the four GEMVs                                  -> r04_fused_launch_phase_trace.log, r04_kernel_phase_trace.log
#   5. A / B of the round's options at the headline shape                                         -> r04_ab_options.log
#   6. dynamic batching 1..32 queries, Mixtral batch 8                                           -> r04_bench_batch.jsonl, r04_bench_mixtral.json
#   8. prefill by prompt length with and without the split-K form of the large-tile GEMM            -> r04_prefill_split_k.log
#      rocprofv3 averages of four 1024-token prefills                                           -> r04_prefill_1024_kernel_stats.csv
#   7. per-kernel rocprofv3 averages of the batched step (2 / 8 / 17 / 32 queries, Mixtral batch 8), rows-GEMM phase trace at 2 queries
#                                                                                                 -> r04_batch*_kernel_stats.csv, r04_rows_gemm_phase_trace.log
set -x
OUT=$PWD/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --prefill-lens "" --batch 0 > $OUT/stats.log 2>&1)
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r04_rocprofv3_kernel_stats.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --prefill-lens "" --batch 0 > $OUT/pmc.log 2>&1)
python tools/pmc_summary.py $(find $OUT/pmc -name "*counter_collection.csv" | head -1) $OUT/r04_pmc_traffic.json > $OUT/pmc_summary.log 2>&1
cp $OUT/r04_pmc_traffic.json profiles/r04_pmc_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r04_bench_n1.json 2> $OUT/bench.err
timeout 600 python bench.py --no-cpu-baseline > $OUT/r04_bench_n1_steps128.json 2>> $OUT/bench.err
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --wdtype q3h --kv-dtype q8 --prefill-lens "" --batch 0 > $OUT/r04_bench_n1_q3h_q8.json 2>> $OUT/bench.err
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --wdtype q3h --prefill-lens "" --batch 0 > $OUT/r04_bench_n1_q3h_f16.json 2>> $OUT/bench.err
timeout 300 python tools/trace_fused.py > $OUT/r04_fused_launch_phase_trace.log 2>&1
timeout 300 python tools/trace_kernels.py > $OUT/r04_kernel_phase_trace.log 2>&1
(for o in fuse_attn attn_kt step_tail fuse_wo fuse_wo_ffn; do timeout 300 python tools/ab_option.py $o --steps 20 --prompt 21 --kernels; done; timeout 300 python tools/ab_option.py graph_steps --values 1,4,8,16 --steps 20 --prompt 21) > $OUT/r04_ab_options.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --shape mixtral_8x7b --batch 8 --steps 64 > $OUT/r04_bench_mixtral.json 2>> $OUT/bench.err
IFA_BATCH_SIZES=1,2,4,8,16,17,24,32 timeout 600 python tools/bench_batch.py > $OUT/r04_bench_batch.jsonl 2>> $OUT/bench.err
bash tools/profile_batch.sh > $OUT/profile_batch.log 2>&1
for f in gpurun_out/prof_batch/*_kernel_stats.csv; do cp $f $OUT/r04_$(basename $f); done
(IFA_NO_GRAPH=1 IFA_ROWS_TRACE=1 timeout 200 python tools/batch_steps.py llama2_7b 2 2 2>&1 | grep "rows-trace" | tail -12; IFA_NO_GRAPH=1 IFA_ROWS_TRACE=1 timeout 200 python tools/batch_steps.py llama2_7b 32 2 2>&1 | grep "rows-trace" | tail -12) > $OUT/r04_rows_gemm_phase_trace.log 2>&1
(for t in 256 512 768 1024 2048; do timeout 200 python tools/prefill_steps.py llama2_7b $t 3 2>&1 | tail -1; IFA_NO_SPLITK=1 timeout 200 python tools/prefill_steps.py llama2_7b $t 3 2>&1 | tail -1 | sed 's/$/   (split-K off)/'; done) > $OUT/r04_prefill_split_k.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pfst -o st -- python $R/tools/prefill_steps.py llama2_7b 1024 4 > $OUT/pfst.log 2>&1)
head -24 $(find $OUT/pfst -name "*kernel_stats.csv" | head -1) > $OUT/r04_prefill_1024_kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc $OUT/pfst
ls -la $OUT
tail -3 $OUT/bench.err
