cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "nibble_formats or operand or fused_batched" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-240 | head
timeout 600 python bench.py --no-cpu-baseline --wdtype q3h --kv-dtype q8 --prefill-lens "" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q3h q8', round(j['value'],1), j.get('batch_decode'))"
