#!/bin/bash
# SQ counters of the two-pass prefill attention kernel (own PMC runs); output gpurun_out/pmc_attn.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_attn
mkdir -p $OUT
export T=${1:-4096} KEYS=0
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $OUT/sq -o sq -- python $OLDPWD/tools/probes/attn_phases.py > $OUT/sq.log 2>&1)
python tools/sq_summary.py $(find $OUT/sq -name "*counter_collection.csv" | head -1) k_attention > gpurun_out/pmc_attn.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -o sq -- python $OLDPWD/tools/probes/attn_phases.py > $OUT/sq2.log 2>&1)
python tools/sq_summary.py $(find $OUT/sq2 -name "*counter_collection.csv" | head -1) k_attention >> gpurun_out/pmc_attn.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS --output-format csv -d $OUT/sq3 -o sq -- python $OLDPWD/tools/probes/attn_phases.py > $OUT/sq3.log 2>&1)
python tools/sq_summary.py $(find $OUT/sq3 -name "*counter_collection.csv" | head -1) k_attention >> gpurun_out/pmc_attn.txt 2>&1
rm -rf $OUT
