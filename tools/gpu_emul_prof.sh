# kernel stats of the emulated rank 0 of an 8-GPU job (collectives replaced by their local part), key-sharded and hybrid
mkdir -p gpurun_out/emul
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for sh in keys hybrid; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/emul/$sh -o s -- python3 $R/bench.py --emulate-rank-of 8 --shard $sh --no-extras --steps 5 --warmup 3 > $R/gpurun_out/emul/$sh.log 2>&1
  f=$(find $R/gpurun_out/emul/$sh -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/summarize_rocprof.py $f $R/gpurun_out/emul/${sh}_kernel_stats.csv
  rm -rf $R/gpurun_out/emul/$sh
  tail -1 $R/gpurun_out/emul/$sh.log | cut -c1-300
done
