mkdir -p gpurun_out/soak_long
{
timeout 700 python tools/soak_filtered.py 560 101 filtered 2>&1 | grep -v amdgpu.ids | tail -2
timeout 700 python tools/soak_filtered.py 560 102 index 2>&1 | grep -v amdgpu.ids | tail -2
timeout 700 python tools/soak_filtered.py 560 103 spec 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python tools/soak_filtered.py 280 104 prior 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python tools/shard_soak.py soak 280 105 2>&1 | grep -v amdgpu.ids | tail -2
} > gpurun_out/soak_long/soak.txt 2>&1
cat gpurun_out/soak_long/soak.txt
