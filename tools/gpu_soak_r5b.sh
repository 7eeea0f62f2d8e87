mkdir -p gpurun_out/i8c
for mode in filtered index prior spec; do
  timeout 400 python tools/soak_filtered.py ${1:-150} 5 $mode 2>&1 | grep -v amdgpu.ids | tail -2
done > gpurun_out/i8c/soak.txt 2>&1
cat gpurun_out/i8c/soak.txt
