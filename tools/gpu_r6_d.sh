#!/bin/bash
R=$(pwd)
O=$R/gpurun_out/r6d
mkdir -p $O
one() {  # tag, args...
  tag=$1; shift
  python bench.py --no-extras --steps 10 --warmup 5 "$@" 2>$O/$tag.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), d['config']['layout'], [(l['launch'], l['dtype'], l['ms'], l.get('candidates_per_query')) for l in d['roofline'].get('levels', [])], d.get('first_bound', {}).get('group'))"
}
for G in 8 4 2; do
  one keys_G$G --emulate-rank-of $G --shard keys
done
for fr in 4 8 "16,3" "8,2"; do
  RAGRAPH_FILTER_FRACS=$fr one "keys_G8_fracs_$fr" --emulate-rank-of 8 --shard keys
done
RAGRAPH_FILTER_SCORED_SHARDS=8 one keys_G8_scored --emulate-rank-of 8 --shard keys
RAGRAPH_FILTER_SCORED_SHARDS=8 RAGRAPH_FILTER_FRACS=8 one keys_G8_scored_fr8 --emulate-rank-of 8 --shard keys
RAGRAPH_FILTER_SCORED_SHARDS=4 one keys_G4_scored --emulate-rank-of 4 --shard keys
one hybrid_G8 --emulate-rank-of 8 --shard hybrid
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/em8 -o s -- python3 $R/bench.py --emulate-rank-of 8 --shard keys --no-extras --steps 5 --warmup 5 > $O/emul_keys_8.log 2>&1
f=$(find $O/em8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/emul_keys_8_kernel_stats.csv; rm -rf $O/em8
