mkdir -p gpurun_out/i8c
for l in 15 30 60 120 240 480; do
  echo "== lambda $l"
  RAGRAPH_I8_CUT_LAMBDA=$l python tools/i8_classes_probe.py gauss 4096 200000 256 2>&1 | grep -v amdgpu.ids | grep "two scales\|levels (\|ms per"
  RAGRAPH_I8_CUT_LAMBDA=$l python tools/i8_classes_probe.py gauss 100000 1000000 256 2>&1 | grep "levels (\|ms per"
  RAGRAPH_I8_CUT_LAMBDA=$l python tools/i8_classes_probe.py gauss 256 4000000 64 2>&1 | grep "two scales\|levels (\|ms per"
done > gpurun_out/i8c/lambda.txt 2>&1
cat gpurun_out/i8c/lambda.txt
