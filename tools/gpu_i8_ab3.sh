#!/bin/bash
# Schedule sweep with the int8 levels in place (one box): level ends as fractions of the bank, first-sample divisor.
R=$(pwd); OUT=$R/gpurun_out/r3_i8_ab3.txt; : > $OUT
run() {
  python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$1: ms_per_step', d['ms_per_step'], 'filter kernels ms', r.get('launch_ms'), 'retrieval call ms', r.get('retrieval_call_ms'))" >> $OUT
}
run "default (fracs 32,4; int8 levels 2)"
for fr in "16,2" "64,8" "32,8" "16,4" "64,4" "16" "8" "32"; do
  export RAGRAPH_FILTER_FRACS=$fr
  for n in 1 2; do RAGRAPH_FILTER_I8=$n run "fracs $fr int8 levels $n"; done
done
unset RAGRAPH_FILTER_FRACS
for nd in 32 128; do RAGRAPH_FILTER_N0DIV=$nd run "n0div $nd"; done
cat $OUT
