#!/bin/bash
# SQ wave-cycle shares of the rows GEMM kernels (own PMC run; output gpurun_out/pmc_rows.txt)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_rows
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $OUT/sq -o sq -- python $OLDPWD/tools/bench_rows.py > $OUT/sq.log 2>&1)
python tools/sq_summary.py $(find $OUT/sq -name "*counter_collection.csv" | head -1) k_gemm_rows > gpurun_out/pmc_rows.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -o sq -- python $OLDPWD/tools/bench_rows.py > $OUT/sq2.log 2>&1)
python tools/sq_summary.py $(find $OUT/sq2 -name "*counter_collection.csv" | head -1) k_gemm_rows >> gpurun_out/pmc_rows.txt 2>&1
rm -rf $OUT
