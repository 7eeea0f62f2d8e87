#!/bin/bash
# Column-sliced SpMM experiment (tools/microbench/spmm_panel_bench.hip) on the GPU box: hop time per variant, then
# FETCH_SIZE / WRITE_SIZE / L2 hit counters per variant in separate PMC passes (kernel-trace only).
#   gpurun --timeout 1200 -- bash tools/gpu_spmm_panel.sh  -> gpurun_out/spmm_panel.txt
R=$(pwd); O=$R/gpurun_out/spmm_panel; mkdir -p $O
OUT=$R/gpurun_out/spmm_panel.txt; : > $OUT
hipcc --offload-arch=gfx950 -O3 -I $R/include -o /tmp/spmm_panel $R/tools/microbench/spmm_panel_bench.hip \
  -L $R/ragraph_amd/csrc -lragraph_hip -Wl,-rpath,$R/ragraph_amd/csrc || exit 1
cd /tmp && export TMPDIR=/tmp
/tmp/spmm_panel all 20 >> $OUT 2>&1
for v in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf $O/p
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p -o s -- /tmp/spmm_panel $v 5 > $O/log.txt 2>&1
    python3 $R/tools/pmc_summary.py "$O/p/**/*counter_collection.csv" spmm 2>&1 | tr -s " " | sed "s/^/  pmc $v: /" >> $OUT
  done
done
rm -rf $O/p
cat $OUT
