# SQ counters of the one-launch encoder (spmm_linear_stream_kernel) inside the GNN forward: separate passes, kernel trace only
R=$(pwd); O=$R/gpurun_out/agglinsq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/sq.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf $O/q$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/q$i -o p -- python3 $R/tools/prof_gnn.py 10 > /dev/null 2>&1
  echo "== pass $i: $set" >> $O/sq.txt
  python3 $R/tools/pmc_summary.py "$O/q$i/**/*counter_collection.csv" spmm_linear_stream >> $O/sq.txt 2>&1
  rm -rf $O/q$i
done
cat $O/sq.txt
