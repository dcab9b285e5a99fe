#!/bin/bash
# A / B of the prologue order of the whole-row staging (rows GEMM, CH == 2: w2 of a 2..4-query step): IFA_ROWS_WHOLE_ORDER 0 / 1 / 2
# built as lib_variants/{wo1,wo2} by `IFA_SWEEP_UNIT=ifa_gemm_rows_mo python tools/sweep_variants.py build wo1=-DIFA_ROWS_WHOLE_ORDER=1 wo2=-DIFA_ROWS_WHOLE_ORDER=2`
OUT=gpurun_out/whole_order; mkdir -p $OUT
for rep in 1 2; do
for v in base wo1 wo2; do
  L=""; [ $v != base ] && L=$PWD/lib_variants/$v/libinferflow_amd.so
  echo "== $v rep $rep" >> $OUT/ab.log
  IFA_LIB=$L IFA_BATCH_SIZES="2,3,4" timeout 300 python tools/bench_batch.py 2>&1 | grep queries >> $OUT/ab.log
done; done
for v in base wo1 wo2; do
  L=""; [ $v != base ] && L=$PWD/lib_variants/$v/libinferflow_amd.so
  echo "== $v" >> $OUT/trace.log
  (IFA_LIB=$L IFA_NO_GRAPH=1 IFA_ROWS_TRACE=1 timeout 200 python tools/batch_steps.py llama2_7b 2 2 2>&1 | grep "rows-trace" | tail -12 | grep -A2 "nblk=344") >> $OUT/trace.log 2>&1
done
cat $OUT/ab.log
