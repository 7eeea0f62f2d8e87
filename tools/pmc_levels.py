#!/usr/bin/env python3
"""Per-DISPATCH counter values (rocprofv3 --pmc counter_collection.csv) of the kernels whose name contains argv[2], in dispatch
order -- the three levels of a retrieval are three launches of one kernel.   python tools/pmc_levels.py "<glob>" topk_filter_kernel"""
import csv, glob, sys
from collections import defaultdict

pat, only = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "topk_filter_kernel"
per = defaultdict(lambda: defaultdict(float))
names = {}
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if only in n:
            d = int(r["Dispatch_Id"])
            per[d][r["Counter_Name"]] += float(r["Counter_Value"])
            names[d] = n.split("(")[0].replace("void ", "")[:60]
for d in sorted(per):
    print(f"dispatch {d:5d} {names[d]:60s} " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(per[d].items())))
