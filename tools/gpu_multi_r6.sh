#!/bin/bash
# Round-5 N > 1 checks that ONE GPU allows: the bench job over gloo with 2 ranks (key-sharded headline + query-sharded) and
# 4 ranks (hybrid 2 x 2 headline + the two pure layouts), the watchdog on a rank hung on purpose, and the single-process
# emulations of rank 0 of 8 for the three layouts.  Output: gpurun_out/<tag>/.
tag=${1:-r6multi}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p $O
S="--steps 3 --warmup 1 --nodes 20000 --bank 200000"
MASTER_PORT=29611 timeout 300 python bench.py --gpus 2 --backend gloo $S > $O/gloo2.json 2> $O/gloo2.err; echo "gloo2 rc=$?"
MASTER_PORT=29612 timeout 400 python bench.py --gpus 4 --backend gloo --shard hybrid $S > $O/gloo4_hybrid.json 2> $O/gloo4_hybrid.err; echo "gloo4 hybrid rc=$?"
RAGRAPH_BENCH_HANG_RANK=1 MASTER_PORT=29613 timeout 200 python bench.py --gpus 2 --backend gloo --dist-timeout 20 $S > $O/hang.json 2> $O/hang.err; echo "hang rc=$? (expected non-zero)"
grep -h "ranks_seen\|bench_watchdog" $O/*.err | cut -c1-600
for sh in keys queries hybrid; do
  timeout 300 python bench.py --emulate-rank-of 8 --shard $sh --steps 10 --warmup 3 --no-extras > $O/emul8_$sh.json 2> $O/emul8_$sh.err
  python - <<PY
import json
d = json.loads(open("$O/emul8_$sh.json").read().strip().splitlines()[-1])
print("emulated rank 0 of 8, $sh:", d["ms_per_step"], "ms/step;", d["config"]["layout"], "rows/gpu", d["config"]["bank_rows_per_gpu"], "queries/gpu", d["config"]["queries_per_gpu"])
PY
done
for f in gloo2 gloo4_hybrid; do python - <<PY
import json
d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", d["n_gpus"], d["config"]["layout"], d["ms_per_step"], {k: (v["ms_per_step"], v["layout"]) for k, v in d.items() if isinstance(v, dict) and "layout" in v and k != "config"}, d.get("ranks_seen"), d.get("verified", {}).get("identical"))
print(json.dumps(d.get("collectives_per_step")))
PY
done
