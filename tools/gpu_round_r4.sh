#!/bin/bash
# Round-4 measurement pass on the GPU box: the bench line, rocprofv3 kernel stats of the same command, the two PMC traffic
# passes (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only), and the small-batch retrieval calls (B = 1, 16, 256)
# with kernel stats + FETCH/WRITE.  Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/.
tag=${1:-r4}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p $O
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras > $O/pmc_write.log 2>&1
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
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python tools/summarize_rocprof.py $f $O/kernel_stats.csv
t=$(find $O/stats -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_summary.py $t > $O/trace_summary.txt
python tools/pmc_summary.py "$O/pmc_fetch/**/*counter_collection.csv" > $O/pmc_fetch.txt 2>&1
python tools/pmc_summary.py "$O/pmc_write/**/*counter_collection.csv" > $O/pmc_write.txt 2>&1
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
cut -c1-600 $O/bench.json
head -8 $O/kernel_stats.csv | cut -c1-220
grep -A2 "topk_filter" $O/pmc_fetch.txt | head -8
grep -A2 "topk_filter" $O/pmc_write.txt | head -8
for B in 1 16 256 512 4096; do grep "ms per call" $O/smallb_B$B.log; head -5 $O/smallb_B${B}_kernel_stats.csv | cut -c1-200; grep -A2 "direct_kernel<256, \(true\|false\), false\|topk_small_kernel" $O/smallb_B${B}_pmc.txt | head -6; done
python tools/reference_bank_bench.py --skip-uncollapsed > $O/refbank.txt 2>&1; tail -2 $O/refbank.txt
