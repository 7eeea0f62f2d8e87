#!/bin/bash
# SQ instruction / cycle counters of the filter launches of one bench step (separate --pmc passes; kernel trace only):
#   gpurun --timeout 1200 -- bash tools/gpu_pmc_levels.sh    -> gpurun_out/r4_pmc_levels.txt
R=$(pwd); O=$R/gpurun_out/pmcl; mkdir -p $O
OUT=$R/gpurun_out/r4_pmc_levels.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/avail.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  ok=""
  for c in $set; do if grep -qx "$c" $O/avail.txt; then ok="$ok $c"; fi; done
  [ -z "$ok" ] && continue
  rm -rf $O/p$i
  rocprofv3 --kernel-trace --pmc $ok --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-extras > $O/log$i.txt 2>&1
  echo "== pass $i:$ok" >> $OUT
  python3 $R/tools/pmc_levels.py "$O/p$i/**/*counter_collection.csv" topk_filter_kernel >> $OUT 2>&1
  rm -rf $O/p$i
done
wc -l $O/avail.txt >> $OUT
cat $OUT
