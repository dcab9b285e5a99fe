#!/bin/bash
# A / B on one box, alternating: default build (early K / V requests) against lib_variants/latekv (-DIFA_QA_EARLY_KV=0)
OUT=gpurun_out/early_kv; mkdir -p $OUT; rm -f $OUT/ab.log
python -m pytest tests/test_gpu_fused_attn.py -x -q 2>&1 | grep -E "passed|failed|error" > $OUT/fused_test.log
for rep in 1 2; do for v in early late; do
  L=""; [ $v = late ] && L=$PWD/lib_variants/latekv/libinferflow_amd.so
  echo "== $v rep $rep" >> $OUT/ab.log
  IFA_LIB=$L timeout 300 python tools/early_kv_ab.py 2>&1 | grep context >> $OUT/ab.log
  [ $rep = 1 ] && { echo "== $v q8" >> $OUT/ab.log; IFA_LIB=$L timeout 300 python tools/early_kv_ab.py q8 2>&1 | grep context >> $OUT/ab.log; }
  echo "== $v bench steps 20 / 128" >> $OUT/ab.log
  IFA_LIB=$L timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --prefill-lens "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['last_tokens'])" >> $OUT/ab.log
  IFA_LIB=$L timeout 300 python bench.py --no-cpu-baseline --steps 128 --warmup 16 --prefill-lens "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['last_tokens'])" >> $OUT/ab.log
done; done
cat $OUT/fused_test.log $OUT/ab.log
