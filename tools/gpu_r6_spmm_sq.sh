#!/bin/bash
R=$(pwd)
O=$R/gpurun_out/r6spmmsq
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -I include -o /tmp/spmm_tiled tools/microbench/spmm_tiled_bench.hip -L ragraph_amd/csrc -lragraph_hip -Wl,-rpath,$PWD/ragraph_amd/csrc 2>/dev/null
cd /tmp && export TMPDIR=/tmp
: > $O/sq.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/q$i -o p -- /tmp/spmm_tiled 200 1310720 3 > /dev/null 2>&1
  echo "== pass $i: $set" >> $O/sq.txt
  python3 $R/tools/pmc_summary.py "$O/q$i/**/*counter_collection.csv" kernel >> $O/sq.txt 2>&1
  rm -rf $O/q$i
done
cat $O/sq.txt | cut -c1-200
