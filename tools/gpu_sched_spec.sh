for B in 2048 4096 16384; do
  for L in 0 1 2; do
    if [ $L = 0 ]; then unset RAGRAPH_FILTER_FORCE_L; else export RAGRAPH_FILTER_FORCE_L=$L; fi
    echo "B=$B force_L=$L: $(python tools/prof_small_batch.py $B 1000000 256 10 30 2>/dev/null | tail -1)"
  done
done
unset RAGRAPH_FILTER_FORCE_L
for fr in "" "4" "8" "16,3" "12,3"; do
  if [ -z "$fr" ]; then unset RAGRAPH_FILTER_FRACS; else export RAGRAPH_FILTER_FRACS="$fr"; fi
  echo "bench FRACS='$fr': $(python bench.py --steps 8 --warmup 4 --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], [(l['launch'], l['keys'], l['ms'], l.get('candidates_per_query')) for l in d['roofline']['levels']])")"
done
