#!/bin/bash
# round-5 tuning run (GPU box): every library under lib_variants/ through the headline bench, the fused launch's phase stamps
set -x
mkdir -p gpurun_out
rm -f gpurun_out/sweep.jsonl
timeout 900 python tools/sweep_variants.py run --steps 64 > gpurun_out/r5_sweep.log 2>&1
timeout 200 python tools/trace_fused.py > gpurun_out/r5_trace_fused.log 2>&1
IFA_LIB=$PWD/lib_variants/nacc1/libinferflow_amd.so timeout 200 python tools/trace_fused.py > gpurun_out/r5_trace_fused_nacc1.log 2>&1
timeout 200 python tools/trace_kernels.py > gpurun_out/r5_trace_kernels.log 2>&1
