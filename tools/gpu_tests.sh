#!/bin/bash
# the whole GPU suite as the driver runs it, with the summary line kept (RCCL banners of the rank tests bury it in a tail)
R=$(pwd)
O=$R/gpurun_out/${1:-tests}
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|error|rc=" $O/pytest_gpu.log | tail -5
