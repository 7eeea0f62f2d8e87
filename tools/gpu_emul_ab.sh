#!/bin/bash
# Key-sharded rank emulation (bench.py --emulate-rank-of G) under schedule variants.  -> gpurun_out/r3_emul_ab.txt
R=$(pwd); OUT=$R/gpurun_out/r3_emul_ab2.txt; : > $OUT
run() { python bench.py --emulate-rank-of $1 --shard keys --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], ' retrieval_call_ms', d['roofline'].get('retrieval_call_ms'), [(l['launch'][:7], l['dtype'], l['keys'], l['ms']) for l in d['roofline'].get('levels', [])])"; }
for G in 2 4 8; do
  echo "G=$G default: $(run $G)" >> $OUT
  for fr in 8 16 4 "16,2"; do
    echo "G=$G FRACS=$fr: $(RAGRAPH_FILTER_FRACS=$fr run $G)" >> $OUT
  done
done
cat $OUT
