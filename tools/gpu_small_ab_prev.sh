# the single-launch kernel, product library against build_ab/lib_prev.so (the library before a change), alternating on one box
for r in 1 2; do
  echo "== new"; python tools/small_prior_ab.py 2>&1 | grep -v amdgpu.ids
  echo "== prev"; RAGRAPH_HIP_SO=build_ab/lib_prev.so python tools/small_prior_ab.py 2>&1 | grep -v amdgpu.ids
done
