#!/bin/bash
# round 6, pass A: the new ingestion entries' tests, the edge fine-tuning step's kernel list (no rocprim:: / hipcub:: kernel may
# remain), the node c2 fine-tuning step's, and the kernel stats of an emulated key-sharded rank of 8 and of 2.
R=$(pwd)
O=$R/gpurun_out/r6a
mkdir -p $O
python -m pytest tests/test_gpu_ingest.py tests/test_gpu_backward.py tests/test_gpu_fewshot_edge.py tests/test_gpu_bank_build.py -x -q 2>&1 | tail -3 > $O/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fe -o s -- python3 $R/tools/prof_finetune_edge.py 2 host > $O/ft_edge.log 2>&1
f=$(find $O/fe -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/ft_edge_kernel_stats.csv; rm -rf $O/fe
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fn -o s -- python3 $R/tools/prof_finetune.py 3 > $O/ft_node.log 2>&1
f=$(find $O/fn -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/ft_node_kernel_stats.csv; rm -rf $O/fn
for G in 8 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/em$G -o s -- python3 $R/bench.py --emulate-rank-of $G --shard keys --no-extras --steps 5 --warmup 3 > $O/emul_keys_$G.log 2>&1
  f=$(find $O/em$G -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $O/emul_keys_${G}_kernel_stats.csv; rm -rf $O/em$G
done
cd $R
python tools/prof_finetune_edge.py 2 device > $O/ft_edge_device.log 2>&1
cat $O/tests.log; tail -2 $O/ft_edge.log $O/ft_edge_device.log $O/ft_node.log | cut -c1-300
grep -il "rocprim\|hipcub\|cub::" $O/*.csv
