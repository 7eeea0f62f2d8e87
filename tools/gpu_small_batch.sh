#!/bin/bash
# rocprofv3 kernel stats of the small-batch retrieval calls (B given as arguments) -> gpurun_out/<tag>/smallb_B<B>.csv
tag=$1; shift
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sb$B -o s -- python3 $R/tools/prof_small_batch.py $B > $O/smallb_B$B.log 2>&1
  f=$(find $O/sb$B -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/smallb_B$B.csv
  grep "ms per call" $O/smallb_B$B.log
  head -12 $O/smallb_B$B.csv
  rm -rf $O/sb$B
done
