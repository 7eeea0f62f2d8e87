for fr in "" "8" "4" "16" "16,3"; do
  if [ -z "$fr" ]; then unset RAGRAPH_FILTER_FRACS; else export RAGRAPH_FILTER_FRACS="$fr"; fi
  for sh in keys hybrid; do
  python bench.py --emulate-rank-of 8 --shard $sh --no-extras --steps 10 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FRACS=\"$fr\" $sh: ms_per_step', d['ms_per_step'], ' call', d['roofline'].get('retrieval_call_ms'), [(l['launch'], l['keys'], l['ms']) for l in d['roofline'].get('levels', [])])"
  done
done
