for i in 1 2; do
python bench.py --emulate-rank-of 8 --shard keys --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keys no-cpu-baseline: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'))"
python bench.py --emulate-rank-of 8 --shard keys --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keys no-extras: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'))"
done
python bench.py --emulate-rank-of 8 --shard hybrid --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hybrid: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'))"
