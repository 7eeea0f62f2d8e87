#!/bin/bash
# Scored candidate lists for the int8 levels (RAGRAPH_FILTER_SCORED=0/1), one box: parity tests first, then the bench line
# and a few batch sizes with both.  -> gpurun_out/r3_scored_ab.txt
R=$(pwd); OUT=$R/gpurun_out/r3_scored_ab.txt; : > $OUT
python -m pytest tests/test_gpu_kernels.py -x -q > $R/gpurun_out/r3_scored_tests.txt 2>&1; tail -3 $R/gpurun_out/r3_scored_tests.txt >> $OUT
for s in 0 1 0 1; do
  echo "bench SCORED=$s: $(RAGRAPH_FILTER_SCORED=$s python bench.py --steps 6 --warmup 2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], [ (l["launch"], l["ms"]) for l in d["roofline"]["levels"]], d["roofline"].get("whole_call"))')" >> $OUT
done
for B in 2048 4096 16384 65536; do
  for s in 0 1; do
    echo "B=$B SCORED=$s: $(RAGRAPH_FILTER_SCORED=$s python tools/prof_small_batch.py $B 2>&1 | grep 'ms per call')" >> $OUT
  done
done
cat $OUT
