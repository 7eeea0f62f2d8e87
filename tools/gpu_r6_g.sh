#!/bin/bash
# the c2 step with the hops on the tiled kernel (default) and on the panel kernels, alternating, same box; then the whole GPU suite
R=$(pwd); O=$R/gpurun_out/r6g; mkdir -p $O
for i in 1 2 3; do
  for t in 1 0; do
    RAGRAPH_SPMM_TILED=$t python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiled=$t run $i: ms_per_step', d['ms_per_step'])"
  done
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
