#!/bin/bash
# round 6, first GPU pass of the chained FFN launch: parity, phase trace, A/B against the separate launches
mkdir -p gpurun_out/r6a
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q > gpurun_out/r6a/test_chain.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6a/test_chain.log
timeout 300 python tools/trace_chain.py > gpurun_out/r6a/trace_chain.log 2>&1; echo "rc $?" >> gpurun_out/r6a/trace_chain.log
timeout 400 python tools/ab_option.py fuse_ffn --values 0,1,2 --rounds 3 --steps 64 > gpurun_out/r6a/ab_fuse_ffn.log 2>&1; echo "rc $?" >> gpurun_out/r6a/ab_fuse_ffn.log
tail -3 gpurun_out/r6a/test_chain.log; cat gpurun_out/r6a/trace_chain.log; cat gpurun_out/r6a/ab_fuse_ffn.log
