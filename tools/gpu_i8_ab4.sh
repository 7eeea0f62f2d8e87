#!/bin/bash
# Mid-sized batches: forced schedules (levels L, first-sample size n0) x int8 levels, one box.  -> gpurun_out/r3_i8_ab4.txt
R=$(pwd); OUT=$R/gpurun_out/r3_i8_ab4.txt; : > $OUT
for B in 512 1024 2048; do
  echo "B=$B default: $(python tools/prof_small_batch.py $B 2>&1 | grep 'ms per call')" >> $OUT
  for L in 1 2 3; do
    for n0 in 16384 65536 131072; do
      for n in 0 1 2; do
        [ $n -gt $L ] && continue
        r=$(RAGRAPH_FILTER_FORCE_L=$L RAGRAPH_FILTER_FORCE_N0=$n0 RAGRAPH_FILTER_I8=$n python tools/prof_small_batch.py $B 2>&1 | grep 'ms per call' | sed 's/.*: //')
        echo "B=$B L=$L n0=$n0 int8=$n: $r" >> $OUT
      done
    done
  done
done
cat $OUT
