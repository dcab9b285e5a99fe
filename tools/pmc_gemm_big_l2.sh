#!/bin/bash
# L2 counters of the large-tile prefill GEMM (own PMC run); output gpurun_out/pmc_gemm_big_l2.txt  (values x 100)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_gemm_big_l2
mkdir -p $OUT
ARGS="${@:-4096 4096 4096}"
(cd /tmp && rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/l2 -o l2 -- python $OLDPWD/tools/probes/gemm_big_one.py $ARGS > $OUT/l2.log 2>&1)
python tools/sq_summary.py $(find $OUT/l2 -name "*counter_collection.csv" | head -1) k_gemm_big > gpurun_out/pmc_gemm_big_l2.txt 2>&1
tail -3 $OUT/l2.log >> gpurun_out/pmc_gemm_big_l2.txt
rm -rf $OUT
