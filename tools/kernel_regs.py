"""Registers and scratch of every kernel in a hipcc --save-temps assembly file (the .amdgpu_metadata block):
  python tools/kernel_regs.py file.s [name-substring]"""
import re, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", "  - .agpr_count:" + blk))
    name = f.get("name", "?")
    if sub in name:
        print(f"{name[:110]:110s} vgpr {f.get('vgpr_count'):>4s} agpr {f.get('agpr_count'):>4s} sgpr {f.get('sgpr_count'):>4s} "
              f"scratch {f.get('private_segment_fixed_size'):>5s} lds {f.get('group_segment_fixed_size')}")
