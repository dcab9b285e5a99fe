#!/bin/bash
# Re-take the PMC traffic summary of the current build (it is tagged with the hash of the kernel sources: any csrc edit makes the
# committed one stale and bench.py reports traffic: null) and the headline bench line.  GPU box, repo root; ~1 minute.
set -x
OUT=$PWD/gpurun_out/prof_r04
mkdir -p $OUT; rm -rf $OUT/pmc
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --prefill-lens "" --batch 0 > $OUT/pmc.log 2>&1)
python tools/pmc_summary.py $(find $OUT/pmc -name "*counter_collection.csv" | head -1) $OUT/r04_pmc_traffic.json > $OUT/pmc_summary.log 2>&1
cp $OUT/r04_pmc_traffic.json profiles/r04_pmc_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r04_bench_n1.json 2> $OUT/bench.err
rm -rf $OUT/pmc
python -c "
import json; d=json.load(open('$OUT/r04_bench_n1.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['prefill_tok_s'])"
