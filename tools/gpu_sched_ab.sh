#!/bin/bash
# Schedule A/B for the bench step with scored lists: int8 level count x level ends.  -> gpurun_out/r3_sched_ab.txt
R=$(pwd); OUT=$R/gpurun_out/r3_sched_ab.txt; : > $OUT
run() { python bench.py --steps 5 --warmup 2 --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], [(l["launch"][:7], l["dtype"], l["keys"], l["ms"]) for l in d["roofline"]["levels"]])'; }
echo "default: $(run)" >> $OUT
for i8 in 2 3; do
  for fr in "32,4" "32,8" "16,4" "64,8" "16,2" "8" "16"; do
    echo "I8=$i8 FRACS=$fr: $(RAGRAPH_FILTER_I8=$i8 RAGRAPH_FILTER_FRACS=$fr run)" >> $OUT
  done
done
for nd in 32 128 256; do
  echo "N0DIV=$nd: $(RAGRAPH_FILTER_N0DIV=$nd run)" >> $OUT
  echo "N0DIV=$nd I8=3: $(RAGRAPH_FILTER_I8=3 RAGRAPH_FILTER_N0DIV=$nd run)" >> $OUT
done
cat $OUT
