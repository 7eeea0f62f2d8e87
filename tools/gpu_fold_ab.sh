# ring kernel's folded thresholds (D = 64 / 128 int8 levels): product library against build_ab/lib_prev.so on one box
for r in 1 2; do
  for cfg in "64 4000000 4096 65536" "128 1000000 4096 16384 100000" "256 1000000 100000"; do
    set -- $cfg; D=$1; N=$2; shift 2
    MID_D=$D MID_N=$N python tools/mid_ab.py $@ 2>&1 | grep -v amdgpu.ids | sed "s/^/D=$D N=$N /"
    MID_D=$D MID_N=$N RAGRAPH_HIP_SO=build_ab/lib_prev.so python tools/mid_ab.py $@ 2>&1 | grep -v amdgpu.ids | sed "s/^/D=$D N=$N /"
  done
done
