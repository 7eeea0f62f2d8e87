#!/bin/bash
# Forced schedules at mid batch sizes with the scored lists in place.  -> gpurun_out/r3_mid_sched.txt
R=$(pwd); OUT=$R/gpurun_out/r3_mid_sched.txt; : > $OUT
for B in 300 512 1024 1536; do
  echo "B=$B default: $(python tools/prof_small_batch.py $B 2>&1 | grep 'ms per call' | sed 's/.*: //')" >> $OUT
  for L in 1 2; do
    for n0 in 32768 65536 131072 262144; do
      r=$(RAGRAPH_FILTER_FORCE_L=$L RAGRAPH_FILTER_FORCE_N0=$n0 python tools/prof_small_batch.py $B 2>&1 | grep 'ms per call' | sed 's/.*: //')
      echo "B=$B L=$L n0=$n0: $r" >> $OUT
    done
  done
done
cat $OUT
