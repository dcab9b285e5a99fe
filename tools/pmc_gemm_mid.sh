#!/bin/bash
# SQ counters of the mid-length prefill GEMM (own PMC runs, no tracing besides --kernel-trace); output gpurun_out/pmc_gemm_mid.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_gemm_mid
R=$PWD
mkdir -p $OUT
run() { (cd /tmp && rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT/$2 -o sq -- python $R/tools/prefill_steps.py llama2_7b ${T:-128} 2 > $OUT/$2.log 2>&1); python tools/sq_summary.py $(find $OUT/$2 -name "*counter_collection.csv" | head -1) k_gemm_mid; }
{
run "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" sq1
run "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" sq2
run "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES" sq3
} > gpurun_out/pmc_gemm_mid.txt 2>&1
rm -rf $OUT
cat gpurun_out/pmc_gemm_mid.txt | cut -c1-400
