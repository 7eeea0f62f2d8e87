#!/bin/bash
# round 6: the graph-tiled hop against the panel kernel -- times in every layout (tools/gnn_probe.py), FETCH / WRITE / L2 hit
# counters of one hop of each (separate passes), the GNN forward (tools/prof_gnn.py) with and without the tiled hops
R=$(pwd)
O=$R/gpurun_out/r6spmm
mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -k "tiled or gnn or propagation" 2>&1 | tail -2
python tools/gnn_probe.py 2>&1 | grep -E "tiled|sliced|3 hops" > $O/gnn_probe.txt; cat $O/gnn_probe.txt
python tools/prof_gnn.py 30 > $O/gnn_fwd_tiled.txt 2>&1; RAGRAPH_SPMM_TILED=0 python tools/prof_gnn.py 30 > $O/gnn_fwd_panel.txt 2>&1; tail -1 $O/gnn_fwd_tiled.txt $O/gnn_fwd_panel.txt
cd /tmp && export TMPDIR=/tmp
: > $O/pmc.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$tag -o p -- python3 $R/tools/prof_spmm_hop.py 5 > $O/hop_$tag.log 2>&1
  python3 $R/tools/pmc_summary.py "$O/p_$tag/**/*counter_collection.csv" spmm >> $O/pmc.txt 2>&1
  rm -rf $O/p_$tag
done
cat $O/pmc.txt; tail -1 $O/hop_FETCH_SIZE.log
