#!/bin/bash
# round 6, session 2: whole GPU suite, then the single-GPU step and the emulated ranks on one box (one-launch encoder, DPP prep)
R=$(pwd)
O=$R/gpurun_out/r6j
mkdir -p $O
bash tools/gpu_tests.sh r6j
one() {  # tag, args...
  tag=$1; shift
  python bench.py --no-extras --steps 10 --warmup 5 "$@" 2>$O/$tag.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), d['config']['layout'], [(l['launch'], l['dtype'], l['ms'], l.get('candidates_per_query')) for l in d['roofline'].get('levels', [])])"
}
{
one single
for G in 8 4 2; do
  one keys_G$G --emulate-rank-of $G --shard keys
done
one queries_G8 --emulate-rank-of 8 --shard queries
one hybrid_G8 --emulate-rank-of 8 --shard hybrid
one single_again
} > $O/emul.txt 2>&1
cat $O/emul.txt
