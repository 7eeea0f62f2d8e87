R=$(pwd); O=$R/gpurun_out/r6s; mkdir -p $O
one() { t=$1; shift; python bench.py --no-extras --steps 10 --warmup 5 "$@" 2>$O/$t.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), d['config']['layout'])"; }
{ one single; one keys_G8 --emulate-rank-of 8 --shard keys; one keys_G4 --emulate-rank-of 4 --shard keys; one keys_G2 --emulate-rank-of 2 --shard keys; one queries_G8 --emulate-rank-of 8 --shard queries; one hybrid_G8 --emulate-rank-of 8 --shard hybrid; one single_again; } > $O/emul.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/em8 -o s -- python3 $R/bench.py --emulate-rank-of 8 --shard keys --no-extras --steps 5 --warmup 5 > $O/emul_keys_8.log 2>&1
f=$(find $O/em8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/emul_keys_8_kernel_stats.csv; rm -rf $O/em8
cat $O/emul.txt
