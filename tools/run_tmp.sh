cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O
timeout 900 python bench.py > $O/r03_bench_n1.json 2> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --wdtype q3h --kv-dtype q8 > $O/r03_bench_n1_q3h_q8.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --wdtype q3h > $O/r03_bench_n1_q3h_f16.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --shape mixtral_8x7b --batch 8 --steps 64 > $O/r03_bench_mixtral.json 2>> $O/bench.err
IFA_BATCH_SIZES=1,2,4,8,16,17,24,32 timeout 600 python tools/bench_batch.py > $O/r03_bench_batch.jsonl 2>> $O/bench.err
python - <<'PY'
import json
for f in ["r03_bench_n1.json","r03_bench_n1_q3h_q8.json","r03_bench_n1_q3h_f16.json","r03_bench_mixtral.json"]:
    j=json.loads(open('gpurun_out/prof_r03/'+f).read().strip().splitlines()[-1])
    r=j.get("roofline") or {}
    print(f, round(j["value"],1), "frac", round(r.get("frac",0) or 0,3), "traffic", r.get("traffic"), round((j.get("batch_decode") or {}).get("tok_s",0)), round(j.get("prefill_tok_s",0)), (j.get("persistent_layer_kernel") or {}).get("tok_s"))
PY
cat $O/r03_bench_batch.jsonl | python -c "
import sys,json
print(' '.join('%d:%d' % (json.loads(l)['queries'], json.loads(l)['aggregate_tok_s']) for l in sys.stdin))"
