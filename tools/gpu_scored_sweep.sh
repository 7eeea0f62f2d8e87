#!/bin/bash
# Batch sizes of the ring kernel with scored lists (and the schedule model's scored terms) off / default.  -> gpurun_out/r3_scored_sweep.txt
R=$(pwd); OUT=$R/gpurun_out/r3_scored_sweep.txt; : > $OUT
for B in 2048 3072 4096 8192 16384 32768 100000; do
  echo "SCORED=0 $(RAGRAPH_FILTER_SCORED=0 python tools/level_times.py $B 2>&1 | tail -1)" >> $OUT
  echo "default  $(python tools/level_times.py $B 2>&1 | tail -1)" >> $OUT
done
cat $OUT
