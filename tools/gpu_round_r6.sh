#!/bin/bash
# Round-6 measurement pass (the round-5 script with the driver-shaped bench line) on the GPU box: the bench line, rocprofv3 kernel stats of the same command, the two PMC traffic
# passes (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only), the small-batch retrieval calls (B = 1, 16, 256, 512,
# 4096) with kernel stats + FETCH/WRITE (with the speculative first bound and, B >= 256, with the bound pass), the SQ counters
# of the 512-query call, the GNN forward's kernel stats + FETCH/WRITE, the other configs.  Everything lands in
# gpurun_out/<tag>/; tools/collect_profiles.py copies what is to be judged into profiles/.
tag=${1:-r6}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python tools/bench_configs.py > $O/configs.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 3 --warmup 3 --no-extras > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 3 --no-extras > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 3 --no-extras > $O/pmc_write.log 2>&1
for B in 1 16 256 512 4096; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sb$B -o s -- python3 $R/tools/prof_small_batch.py $B > $O/smallb_B$B.log 2>&1
  f=$(find $O/sb$B -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/smallb_B${B}_kernel_stats.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/sbf$B -o p -- python3 $R/tools/prof_small_batch.py $B 1000000 256 10 10 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/sbw$B -o p -- python3 $R/tools/prof_small_batch.py $B 1000000 256 10 10 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py "$O/sbf$B/**/*counter_collection.csv" > $O/smallb_B${B}_pmc.txt 2>&1
  python3 $R/tools/pmc_summary.py "$O/sbw$B/**/*counter_collection.csv" >> $O/smallb_B${B}_pmc.txt 2>&1
  rm -rf $O/sb$B $O/sbf$B $O/sbw$B
done
# the same calls WITH their bound pass (no speculative first bound): kernel stats only
for B in 256 512 4096; do
  export RAGRAPH_NO_PRIOR=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sbn$B -o s -- python3 $R/tools/prof_small_batch.py $B > $O/smallb_noprior_B$B.log 2>&1
  unset RAGRAPH_NO_PRIOR
  f=$(find $O/sbn$B -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/smallb_noprior_B${B}_kernel_stats.csv
  rm -rf $O/sbn$B
done
# SQ counters of the 512-query call (what the 125-us int8 launch waits on): separate passes, kernel trace only
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/avail.txt
: > $O/pmc_B512_sq.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
  i=$((i+1))
  ok=""
  for c in $set; do if grep -qx "$c" $O/avail.txt; then ok="$ok $c"; fi; done
  [ -z "$ok" ] && continue
  rm -rf $O/q$i
  rocprofv3 --kernel-trace --pmc $ok --output-format csv -d $O/q$i -o p -- python3 $R/tools/prof_small_batch.py 512 1000000 256 10 10 > /dev/null 2>&1
  echo "== pass $i:$ok" >> $O/pmc_B512_sq.txt
  python3 $R/tools/pmc_summary.py "$O/q$i/**/*counter_collection.csv" topk_filter_kernel >> $O/pmc_B512_sq.txt 2>&1
  rm -rf $O/q$i
done
# GNN forward: kernel stats + FETCH / WRITE, c2's graph and the community graph after the automatic reordering
for mode in plain structured; do
  arg=""; [ $mode = structured ] && arg="structured"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/g$mode -o s -- python3 $R/tools/prof_gnn.py 20 $arg > $O/gnn_$mode.log 2>&1
  f=$(find $O/g$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/gnn_${mode}_kernel_stats.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/gf$mode -o p -- python3 $R/tools/prof_gnn.py 10 $arg > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/gw$mode -o p -- python3 $R/tools/prof_gnn.py 10 $arg > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py "$O/gf$mode/**/*counter_collection.csv" > $O/gnn_${mode}_pmc.txt 2>&1
  python3 $R/tools/pmc_summary.py "$O/gw$mode/**/*counter_collection.csv" >> $O/gnn_${mode}_pmc.txt 2>&1
  rm -rf $O/g$mode $O/gf$mode $O/gw$mode
done
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python tools/summarize_rocprof.py $f $O/kernel_stats.csv
t=$(find $O/stats -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_summary.py $t > $O/trace_summary.txt
python tools/pmc_summary.py "$O/pmc_fetch/**/*counter_collection.csv" > $O/pmc_fetch.txt 2>&1
python tools/pmc_summary.py "$O/pmc_write/**/*counter_collection.csv" >> $O/pmc_write.txt 2>&1
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
for i in 1 2; do RAGRAPH_FILTER_I8_DIRECT_D64=0 python tools/d64_ab.py 4000000 spec 2>/dev/null; RAGRAPH_FILTER_I8_DIRECT_D64=1 python tools/d64_ab.py 4000000 spec 2>/dev/null; done > $O/d64_ab.txt
cut -c1-600 $O/bench.json
head -8 $O/kernel_stats.csv | cut -c1-220
for B in 1 16 256 512 4096; do grep "ms per call" $O/smallb_B$B.log; head -5 $O/smallb_B${B}_kernel_stats.csv | cut -c1-200; done
cat $O/pmc_B512_sq.txt | cut -c1-300 | head -40
cat $O/gnn_plain.log $O/gnn_structured.log | grep gnn_forward
