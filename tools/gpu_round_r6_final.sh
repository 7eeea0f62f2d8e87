#!/bin/bash
# Round 6, last pass (after tools/gpu_round_r6.sh + tools/collect_profiles.py have refreshed profiles/r6_pmc_traffic.json): the
# driver-shaped bench line on the final library, the emulated ranks next to the single-GPU step on the same box, and the kernel
# stats of the fine-tuning steps and of an emulated key-sharded rank of 8.
tag=${1:-r6m}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
one() {  # tag, args...
  t=$1; shift
  python bench.py --no-extras --steps 10 --warmup 5 "$@" 2>$O/$t.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), d['config']['layout'], [(l['launch'], l['dtype'], l['ms'], l.get('candidates_per_query')) for l in d['roofline'].get('levels', [])])"
}
{
one single
for G in 8 4 2; do one keys_G$G --emulate-rank-of $G --shard keys; done
for G in 8 4 2; do one queries_G$G --emulate-rank-of $G --shard queries; done
one hybrid_G8 --emulate-rank-of 8 --shard hybrid
one hybrid_G4 --emulate-rank-of 4 --shard hybrid
one single_again
} > $O/emul.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fe -o s -- python3 $R/tools/prof_finetune_edge.py 2 host > $O/ft_edge.log 2>&1
f=$(find $O/fe -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/ft_edge_kernel_stats.csv; rm -rf $O/fe
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fn -o s -- python3 $R/tools/prof_finetune.py 3 > $O/ft_node.log 2>&1
f=$(find $O/fn -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/ft_node_kernel_stats.csv; rm -rf $O/fn
rocprofv3 --kernel-trace --stats --output-format csv -d $O/em8 -o s -- python3 $R/bench.py --emulate-rank-of 8 --shard keys --no-extras --steps 5 --warmup 5 > $O/emul_keys_8.log 2>&1
f=$(find $O/em8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/emul_keys_8_kernel_stats.csv; rm -rf $O/em8
cd $R
cut -c1-300 $O/bench.json; cat $O/emul.txt; for f in $O/ft_node.log $O/ft_edge.log; do tail -n 1 $f | cut -c1-200; done
grep -il "rocprim\|hipcub\|cub::" $O/ft_*.csv || true
