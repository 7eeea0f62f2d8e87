#!/bin/bash
# A/B on ONE box: int8 levels x queries per wave at the bench shape, and int8 levels at mid-sized batches.
R=$(pwd); OUT=$R/gpurun_out/r3_i8_ab2.txt; : > $OUT
for cfg in "1 64" "2 64" "1 128" "2 128"; do
  set -- $cfg
  RAGRAPH_FILTER_I8=$1 RAGRAPH_FILTER_I8_QW=$2 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('int8 levels $1 QW $2: ms_per_step', d['ms_per_step'], 'filter kernels ms', r.get('launch_ms'), 'retrieval call ms', r.get('retrieval_call_ms'))" >> $OUT
done
for B in 512 1024 2048 4096 8192 16384; do
  for n in 0 1 2; do
    echo "B=$B int8 levels $n: $(RAGRAPH_FILTER_I8=$n python tools/prof_small_batch.py $B 2>&1 | grep 'ms per call')" >> $OUT
  done
done
cat $OUT
