#!/bin/bash
# A/B on ONE box: the bench step with 0 / 1 / 2 trailing filter levels on the int8 copy (RAGRAPH_FILTER_I8), then the
# kernel breakdown of the default.   gpurun -- bash tools/gpu_i8_ab.sh   -> gpurun_out/r3_i8_ab.txt
R=$(pwd); OUT=$R/gpurun_out/r3_i8_ab.txt; : > $OUT
for n in 0 1 2 3; do
  RAGRAPH_FILTER_I8=$n python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('int8 levels $n: ms_per_step', d['ms_per_step'], 'queries/s', d['value'], 'filter kernels ms', r.get('launch_ms'), 'retrieval call ms', r.get('retrieval_call_ms'))" >> $OUT
done
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/i8p; rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > $O.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
echo "== kernels, default rule (4 forwards)" >> $OUT
python3 $R/tools/kstats_brief.py $f >> $OUT
rm -rf $O
cat $OUT
