set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_q4.json 2> gpurun_out/r03/bench_q4.err
timeout 300 python bench.py --no-cpu-baseline --wdtype q3h --kv-dtype q8 > gpurun_out/r03/bench_q3h_q8.json 2> gpurun_out/r03/bench_q3h_q8.err
timeout 300 python bench.py --no-cpu-baseline --wdtype q3h > gpurun_out/r03/bench_q3h_f16.json 2> gpurun_out/r03/bench_q3h_f16.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r03/prof_q4 -o q4 --output-format csv -- python bench.py --no-cpu-baseline --steps 64 --warmup 8 > gpurun_out/r03/prof_q4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r03/prof_q3h -o q3h --output-format csv -- python bench.py --no-cpu-baseline --steps 64 --warmup 8 --wdtype q3h --kv-dtype q8 > gpurun_out/r03/prof_q3h.log 2>&1
find gpurun_out/r03 -name "*kernel_trace*" -delete
ls -R gpurun_out/r03 | head -30
cat gpurun_out/r03/bench_q4.json gpurun_out/r03/bench_q3h_q8.json gpurun_out/r03/bench_q3h_f16.json | cut -c1-1500
