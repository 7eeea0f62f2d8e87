for sh in keys queries hybrid; do
python bench.py --emulate-rank-of 8 --shard $sh --no-extras --steps 10 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sh: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), d['config']['layout'], [(l['launch'], l['ms']) for l in d['roofline'].get('levels', [])])"
done
for G in 2 4; do
python bench.py --emulate-rank-of $G --shard queries --no-extras --steps 10 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queries G=$G: ms_per_step', d['ms_per_step'])"
python bench.py --emulate-rank-of $G --shard keys --no-extras --steps 10 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keys G=$G: ms_per_step', d['ms_per_step'])"
done
python bench.py --no-extras --steps 10 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single GPU: ms_per_step', d['ms_per_step'])"
