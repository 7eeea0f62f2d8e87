# kernel chain of the Cora-shaped forward (c1): rocprofv3 kernel stats of tools/prof_c1.py
R=$(pwd); O=$R/gpurun_out/c1chain; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python3 $R/tools/prof_c1.py > $O/log.txt 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1); head -30 $f | cut -c1-170
