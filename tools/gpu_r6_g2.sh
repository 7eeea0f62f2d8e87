#!/bin/bash
# the one-launch encoder: parity, then its halves alone (diagnostic builds)
R=$(pwd)
O=$R/gpurun_out/r6g
mkdir -p $O
rm -f $O/ablate.txt
timeout 120 python -m pytest tests/test_gpu_kernels.py -x -q -k "spmm_linear_one" 2>&1 | tail -2 >> $O/ablate.txt
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export RAGRAPH_HIP_SO=$R/build_ab/lib_$v.so; fi
  echo "== ${v:-product}" >> $O/ablate.txt
  timeout 120 python tools/spmm_linear_probe.py 2>&1 | grep -v amdgpu.ids >> $O/ablate.txt
done
cat $O/ablate.txt
