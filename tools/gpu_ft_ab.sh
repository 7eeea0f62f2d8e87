#!/bin/bash
R=$(pwd)
O=$R/gpurun_out/r6ft
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export RAGRAPH_SPMM_LINEAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$v -o s -- python3 $R/tools/prof_finetune_c2.py 8 > $O/ft_$v.log 2>&1
  f=$(find $O/p$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/ft_${v}_kernel_stats.csv
  t=$(find $O/p$v -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && cp $t $O/ft_${v}_trace.csv
  rm -rf $O/p$v
  grep "fine-tuning" $O/ft_$v.log
done
ls -la $O
