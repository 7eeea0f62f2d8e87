#!/bin/bash
# One GPU-box pass: full GPU test suite, smoke, bench line, rocprofv3 kernel stats and the two PMC traffic passes of the
# same bench command.  Run through gpurun from the repo root; everything lands in gpurun_out/<tag>/.
tag=${1:-round}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p $O
if [ -z "$PROFILE_ONLY" ]; then
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py --small-batch > $O/bench.json 2> $O/bench.err
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python tools/summarize_rocprof.py $f $O/kernel_stats.csv
t=$(find $O/stats -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_summary.py $t > $O/trace_summary.txt
python tools/pmc_summary.py "$O/pmc_fetch/**/*counter_collection.csv" > $O/pmc_fetch.txt 2>&1
python tools/pmc_summary.py "$O/pmc_write/**/*counter_collection.csv" > $O/pmc_write.txt 2>&1
# keep the merged-back directory small
find $O -name "*.csv" -size +2M -delete
cat $O/pytest.txt $O/smoke.txt $O/bench.json 2>/dev/null
tail -3 $O/stats.log
head -4 $O/trace_summary.txt
head -5 $O/kernel_stats.csv
grep -A2 "topk_filter\|topk_stream" $O/pmc_fetch.txt | head -12
grep -A2 "topk_filter\|topk_stream" $O/pmc_write.txt | head -12
RAGRAPH_FORCE_DIST=1 python bench.py --no-cpu-baseline 2>/dev/null | grep metric | cut -c1-260 > $O/bench_forced_dist.txt; cat $O/bench_forced_dist.txt
python bench.py --exact-fp32 --no-cpu-baseline 2>/dev/null | grep metric > $O/bench_exact_fp32.json; cut -c1-200 $O/bench_exact_fp32.json
