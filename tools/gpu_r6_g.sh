#!/bin/bash
# round 6, session 2: the one-launch aggregate-first encoder (ragraph_spmm_linear_f32): parity + A/B
R=$(pwd)
O=$R/gpurun_out/r6g
mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_kernels.py -x -q -k "spmm_linear_one" 2>&1 | tail -5 > $O/tests.log
timeout 300 python tools/spmm_linear_probe.py > $O/probe.txt 2>&1
for v in 1 0; do
  RAGRAPH_SPMM_LINEAR=$v timeout 300 python tools/prof_gnn.py 50 2>&1 | tail -2 > $O/gnn_$v.txt
done
cat $O/tests.log $O/probe.txt $O/gnn_1.txt $O/gnn_0.txt
