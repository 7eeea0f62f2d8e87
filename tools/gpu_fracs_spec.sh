# the 100 000-query step's level ends (RAGRAPH_FILTER_FRACS = "a,b": ends at N/a, N/b) under the speculative first bound (KeyIndex)
for f in "32,4" "64,8" "16,4" "32,8" "64,4" "24,3" "48,6" "128,8" "32,4"; do
  RAGRAPH_FILTER_FRACS=$f python tools/mid_ab.py 100000 2>&1 | grep -v amdgpu.ids
done
