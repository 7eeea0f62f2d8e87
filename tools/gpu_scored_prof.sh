#!/bin/bash
# Kernel durations of the bench step with scored lists off / on (+ the kernel parity tests).  -> gpurun_out/r3_scored_prof.txt
R=$(pwd); OUT=$R/gpurun_out/r3_scored_prof.txt; : > $OUT
python -m pytest tests/test_gpu_kernels.py -x -q > $R/gpurun_out/r3_scored_tests.txt 2>&1; tail -3 $R/gpurun_out/r3_scored_tests.txt >> $OUT
cd /tmp; export TMPDIR=/tmp
for s in 0 1; do
  rm -rf /tmp/sc$s
  RAGRAPH_FILTER_SCORED=$s rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sc$s -o sc -- python3 $R/bench.py --steps 4 --warmup 2 --no-extras > /tmp/sc$s.log 2>&1
  f=$(find /tmp/sc$s -name '*kernel_stats.csv' | head -1)
  echo "== SCORED=$s" >> $OUT; tail -1 /tmp/sc$s.log | cut -c1-200 >> $OUT
  python3 $R/tools/kstats_brief.py $f | head -8 >> $OUT
done
cd $R
for s in 0 1 0 1; do
  echo "bench SCORED=$s: $(RAGRAPH_FILTER_SCORED=$s python bench.py --steps 6 --warmup 2 --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" >> $OUT
done
cat $OUT
