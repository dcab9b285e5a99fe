cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "long_prompt_large_tile" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-240 | head
timeout 600 python bench.py --no-cpu-baseline --wdtype q3h --kv-dtype q8 --batch 0 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q3h q8', round(j['value'],1), j.get('prefill_tok_s_by_prompt_len'))"
