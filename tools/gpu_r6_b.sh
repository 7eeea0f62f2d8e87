#!/bin/bash
# round 6, pass B: the sharded speculative first bound -- the two-rank / four-rank tests on the real library, then the emulated
# rank 0 of G = 2 / 4 / 8 (key-sharded) with and without the group's prior (RAGRAPH_SPEC=0), the other layouts, one GPU.
R=$(pwd)
O=$R/gpurun_out/r6b
mkdir -p $O
python -m pytest tests/test_gpu_two_rank.py tests/test_gpu_configs.py -x -q 2>&1 | tail -15 > $O/tests.log
one() {  # tag, args...
  tag=$1; shift
  python bench.py --no-extras --steps 10 --warmup 5 "$@" 2>$O/$tag.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), d['config']['layout'], [(l['launch'], l['dtype'], l['ms']) for l in d['roofline'].get('levels', [])], d.get('first_bound', {}).get('group'))"
}
for G in 8 4 2; do
  one keys_G$G --emulate-rank-of $G --shard keys
  RAGRAPH_SPEC=0 one keys_G${G}_nospec --emulate-rank-of $G --shard keys
done
RAGRAPH_FILTER_SCORED_SHARDS=0 one keys_G2_unscored --emulate-rank-of 2 --shard keys
RAGRAPH_FILTER_SCORED_SHARDS=4 one keys_G4_scored --emulate-rank-of 4 --shard keys
one hybrid_G8 --emulate-rank-of 8 --shard hybrid
one queries_G8 --emulate-rank-of 8 --shard queries
one single
cat $O/tests.log | tail -5
