cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_host_engine.py tests/test_gpu_comm.py -q -m gpu -k "batch or dynamic or operand" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-200 | head
IFA_BATCH_SIZES=2,8,16,32 timeout 300 python tools/bench_batch.py 2>/dev/null
