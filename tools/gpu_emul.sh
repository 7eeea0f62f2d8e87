#!/bin/bash
# One rank of a G-GPU job emulated on ONE GPU (bench.py --emulate-rank-of G: the rank's shard of the work, collectives
# replaced by their local cost) for both layouts, and the per-kernel breakdown of the key-sharded rank (rocprofv3 kernel
# stats).  An estimate for DESIGN.md section 5, never a bench line.     gpurun --timeout 1500 -- bash tools/gpu_emul.sh 8
#   -> gpurun_out/r3_emul.txt   (copy to profiles/)
R=$(pwd); O=$R/gpurun_out/emul; mkdir -p $O
OUT=$R/gpurun_out/${EMUL_TAG:-r4}_emul.txt; : > $OUT
for G in 2 4 8; do
  for shard in keys queries; do
    python bench.py --emulate-rank-of $G --shard $shard --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('emulated rank of $G, $shard-sharded: ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'))" >> $OUT
  done
done
cd /tmp && export TMPDIR=/tmp
for G in "$@"; do
  rm -rf $O/p
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python3 $R/bench.py --emulate-rank-of $G --steps 3 --warmup 1 --no-extras > $O/log$G.txt 2>&1
  f=$(find $O/p -name "*kernel_stats.csv" | head -1)
  echo "== kernels of the key-sharded rank of $G (4 forwards)" >> $OUT
  python3 - $f >> $OUT <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
tot = sum(float(r[2]) for r in rows)
for r in rows[:14]:
    n = r[0].split("(")[0].replace("void ", "")[:70]
    print(f"   {n:70s} calls={r[1]:>5s} total_ms={float(r[2]) / 1e6:8.2f} avg_us={float(r[3]) / 1e3:9.1f}")
print("   sum of all kernels, ms:", round(tot / 1e6, 2))
PY
  rm -rf $O/p
done
cat $OUT
