#!/bin/bash
# A/B of library builds on ONE box: per variant (build_ab/lib_<v>.so, or "base" = the in-tree library) the call time of
# tools/prof_small_batch.py at the given batch sizes and the rocprofv3 kernel averages of the filter kernels.
#   gpurun -- bash tools/gpu_ab_libs.sh "256 128" base la8 noepi      -> gpurun_out/ab_libs.txt
R=$(pwd); O=$R/gpurun_out/ab; mkdir -p $O
OUT=$R/gpurun_out/ab_libs.txt; : > $OUT
BS=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset RAGRAPH_HIP_SO; else export RAGRAPH_HIP_SO=$R/build_ab/lib_$v.so; fi
  for B in $BS; do
    rm -rf $O/s
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o s -- python3 $R/tools/prof_small_batch.py $B > $O/log.txt 2>&1
    echo "== $v B=$B: $(grep 'ms per call' $O/log.txt)" >> $OUT
    f=$(find $O/s -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python3 $R/tools/kstats_brief.py $f >> $OUT
  done
done
rm -rf $O/s
cat $OUT
