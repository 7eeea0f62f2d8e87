#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r6h; mkdir -p $O
python -m pytest tests/test_gpu_two_rank.py tests/test_gpu_models.py tests/test_gpu_configs.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
one() {  # tag, args...
  tag=$1; shift
  python bench.py --no-extras --steps 10 --warmup 5 "$@" 2>$O/$tag.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), d['config']['layout'])"
}
one single
for G in 8 4 2; do
  one queries_G$G --emulate-rank-of $G --shard queries
done
one hybrid_G8 --emulate-rank-of 8 --shard hybrid
one hybrid_G4 --emulate-rank-of 4 --shard hybrid
one keys_G8 --emulate-rank-of 8 --shard keys
