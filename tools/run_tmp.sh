cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; mkdir -p gpurun_out/r03
IFA_BATCH_SIZES=32 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r03/prof_b32 -o b32 --output-format csv -- python tools/bench_batch.py > gpurun_out/r03/prof_b32.log 2>&1
find gpurun_out/r03/prof_b32 -name "*kernel_trace*" -delete
