# mid batches through KeyIndex, product library against build_ab/lib_prev.so, alternating on one box
for r in 1 2 3; do
  python tools/mid_ab.py 300 512 1100 2048 4096 16384 2>&1 | grep -v amdgpu.ids
  RAGRAPH_HIP_SO=build_ab/lib_prev.so python tools/mid_ab.py 300 512 1100 2048 4096 16384 2>&1 | grep -v amdgpu.ids
done
