#!/bin/bash
# A/B on ONE box: query groups per wave of the int8 levels (RAGRAPH_FILTER_I8_QW = 64 / 96 / 128) at the bench shape.
R=$(pwd); OUT=$R/gpurun_out/r3_i8_qw.txt; : > $OUT
for rep in 1 2; do
for qw in 64 96 128; do
  RAGRAPH_FILTER_I8_QW=$qw python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('int8 QW $qw: ms_per_step', d['ms_per_step'], 'levels', [(l['dtype'], l['ms']) for l in r['levels']], 'retrieval call ms', r.get('retrieval_call_ms'))" >> $OUT
done
done
RAGRAPH_FILTER_I8_QW=96 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -k "int8 or fullsize or full_size" 2>&1 | tail -2 >> $OUT
cat $OUT
