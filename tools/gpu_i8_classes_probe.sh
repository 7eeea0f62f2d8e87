mkdir -p gpurun_out/i8c
for kind in gauss clustered onehot; do
  python tools/i8_classes_probe.py $kind 4096 200000 256
  RAGRAPH_I8_ONE_SCALE=1 python tools/i8_classes_probe.py $kind 4096 200000 256
done > gpurun_out/i8c/probe.txt 2>&1
python tools/i8_classes_probe.py gauss 100000 1000000 256 >> gpurun_out/i8c/probe.txt 2>&1
RAGRAPH_I8_ONE_SCALE=1 python tools/i8_classes_probe.py gauss 100000 1000000 256 >> gpurun_out/i8c/probe.txt 2>&1
cat gpurun_out/i8c/probe.txt
