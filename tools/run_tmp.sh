cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench_line.py tests/test_gpu_oneshot_ipc.py tests/test_gpu_comm.py -q -m gpu 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | head -20
