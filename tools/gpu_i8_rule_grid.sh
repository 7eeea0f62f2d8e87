mkdir -p gpurun_out/i8c
for f in ${CANDFS:-3.0 2.0 1.5}; do
  RAGRAPH_FILTER_I8_CANDF=$f python tools/i8_rule_grid.py 2>&1 | grep -v amdgpu.ids > gpurun_out/i8c/grid_$f.txt
done
ls gpurun_out/i8c
