#!/bin/bash
# SpMM locality experiment on the GPU box: hop time + L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes) per
# node ordering and per XCD run length (RAGRAPH_SPMM_XCD_RUN: consecutive workgroups = 4 rows each at D = 256 that share an XCD).
#   gpurun --timeout 1500 -- bash tools/gpu_spmm_locality.sh       -> gpurun_out/spmm_locality.txt
R=$(pwd); O=$R/gpurun_out/spmm_loc; mkdir -p $O
OUT=$R/gpurun_out/spmm_locality.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
for run in 32 256 1024; do
  export RAGRAPH_SPMM_XCD_RUN=$run
  for mode in random shuffled rcm truth; do
    python3 $R/tools/spmm_locality.py $mode 2>&1 | grep "mode=\|reordered" | sed "s/^/xcd_run=$run /" >> $OUT
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $O/p
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p -o s -- python3 $R/tools/spmm_locality.py $mode 20 > $O/log.txt 2>&1
      python3 $R/tools/pmc_summary.py "$O/p/**/*counter_collection.csv" 2>&1 | grep -A1 "spmm_" | grep -v "^--" | tr -s " " | paste - - | sed "s/^/  pmc xcd_run=$run $mode: /" >> $OUT
    done
  done
done
rm -rf $O/p
cat $OUT
