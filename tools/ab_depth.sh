#!/bin/bash
# same-box A/B of the tile kernel's plan depth on c2-like shapes (+ fabric-side traffic of each)
R=$(pwd); O=$R/gpurun_out/ab_depth; mkdir -p $O
for d in 0 1 9; do
  echo "== RAGRAPH_TOPK_DEPTH=$d"
  RAGRAPH_TOPK_DEPTH=$d python tools/quick_topk_bench.py 100000,1000000,256,10 50000,1000000,256,10 20000,1000000,128,10 2>&1 | grep "B="
done
cd /tmp && export TMPDIR=/tmp
for d in 0 9; do
  export RAGRAPH_TOPK_DEPTH=$d
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/F$d -- python3 $R/tools/quick_topk_bench.py 100000,1000000,256,10 > /dev/null 2>&1
  echo "== FETCH_SIZE depth cap $d"; python3 $R/tools/pmc_summary.py "$O/F$d/**/*counter_collection.csv" topk_stream
done
find $O -name "*.csv" -size +1M -delete
