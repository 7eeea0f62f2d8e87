# ring kernel at D = 64: the SIMD partners' lead (RAGRAPH_FILTER_PARTNER_LEAD, read per call) and the groups per wave
for lead in 1 0 2 3 4 1; do
  RAGRAPH_FILTER_PARTNER_LEAD=$lead MID_D=64 MID_N=4000000 python tools/mid_ab.py 4096 65536 2>&1 | grep -v amdgpu.ids
done
for qw in 64 96 128; do
  RAGRAPH_FILTER_I8_QW=$qw MID_D=64 MID_N=4000000 python tools/mid_ab.py 4096 65536 2>&1 | grep -v amdgpu.ids
done
