#!/bin/bash
R=$(pwd)
O=$R/gpurun_out/r6e
mkdir -p $O
python -m pytest tests/test_gpu_two_rank.py -x -q 2>&1 | tail -3 > $O/tests.log
one() {  # tag, args...
  tag=$1; shift
  python bench.py --no-extras --steps 10 --warmup 5 "$@" 2>$O/$tag.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), d['config']['layout'], [(l['launch'], l['dtype'], l['ms'], l.get('candidates_per_query')) for l in d['roofline'].get('levels', [])])"
}
for G in 8 4 2; do
  for two in 0 2; do
    RAGRAPH_FILTER_SPEC_SHARDS_TWO_LEVELS=$two one keys_G${G}_two$two --emulate-rank-of $G --shard keys
  done
done
for fr in 3 5 6; do
  RAGRAPH_FILTER_SPEC_SHARDS_TWO_LEVELS=0 RAGRAPH_FILTER_FRACS=$fr one "keys_G8_fracs_$fr" --emulate-rank-of 8 --shard keys
done
RAGRAPH_FILTER_SCORED_SHARDS=4 one keys_G4_scored --emulate-rank-of 4 --shard keys
RAGRAPH_FILTER_SCORED_SHARDS=4 RAGRAPH_FILTER_SPEC_SHARDS_TWO_LEVELS=0 one keys_G4_scored_three --emulate-rank-of 4 --shard keys
one hybrid_G8 --emulate-rank-of 8 --shard hybrid
one single
cat $O/tests.log
