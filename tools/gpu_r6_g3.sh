#!/bin/bash
# whole GPU suite on the one-launch encoder + the GNN forward both ways
R=$(pwd)
O=$R/gpurun_out/r6g
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/suite.log
for v in 1 0; do
  RAGRAPH_SPMM_LINEAR=$v timeout 300 python tools/prof_gnn.py 50 2>&1 | tail -1 > $O/gnn_$v.txt
done
cat $O/suite.log $O/gnn_1.txt $O/gnn_0.txt
