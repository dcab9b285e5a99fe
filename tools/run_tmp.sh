cd $GRAFT_REPO_ROOT
for a in "" "--wdtype q3h --kv-dtype q8" "--kv-dtype q8"; do
timeout 300 python bench.py --no-cpu-baseline --prefill-lens "" --batch 0 $a 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['config']['workload'][:60], round(j['value'],1), round(j['kernels']['attn']['us'],2))"
done
