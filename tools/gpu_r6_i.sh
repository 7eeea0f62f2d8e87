#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r6i; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_backward.py tests/test_gpu_two_rank.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
one() {  # tag, args...
  tag=$1; shift
  python bench.py --no-extras --steps 10 --warmup 5 "$@" 2>$O/$tag.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), d['config']['layout'])"
}
one single
for sh in keys queries hybrid; do
  for G in 8 4 2; do
    [ $sh = hybrid ] && [ $G = 2 ] && continue
    one ${sh}_G$G --emulate-rank-of $G --shard $sh
  done
done
