#!/usr/bin/env python3
"""Per-kernel duration summary (median/min/max, grid, VGPRs, LDS) from a rocprofv3 *_kernel_trace.csv."""
import csv, glob, statistics, sys
from collections import defaultdict

f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/**/*kernel_trace.csv", recursive=True))[-1]
only = sys.argv[2] if len(sys.argv) > 2 else "ragraph"
d = defaultdict(list)
meta = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if only in n:
        short = n.split("(")[0].replace("void ", "")
        key = (short, r["Grid_Size_X"])
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[key] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:60s} grid={k[1]:>9s} n={len(v):3d} med={statistics.median(v)/1e3:10.1f}us min={min(v)/1e3:10.1f} max={max(v)/1e3:10.1f} vgpr={meta[k][0]} lds={meta[k][1]} scratch={meta[k][2]}")
