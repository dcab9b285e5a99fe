#!/bin/bash
# SQ counters of the prefill GEMM kernels (GPU box): bash tools/pmc_gemm.sh  -> gpurun_out/pmc_gemm.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -o a -- python $OLDPWD/tools/bench_gemm_big.py > $OUT/a.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -o b -- python $OLDPWD/tools/bench_gemm_big.py > $OUT/b.log 2>&1)
python tools/sq_summary.py $(find $OUT/a -name "*counter_collection.csv" | head -1) k_gemm > gpurun_out/pmc_gemm.txt 2>&1
python tools/sq_summary.py $(find $OUT/b -name "*counter_collection.csv" | head -1) k_gemm >> gpurun_out/pmc_gemm.txt 2>&1
rm -rf $OUT
