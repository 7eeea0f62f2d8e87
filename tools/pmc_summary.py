#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per (kernel, counter): mean per dispatch."""
import csv, glob, sys
from collections import defaultdict

pat = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else "ragraph"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if only in n:
            short = n.split("(")[0].replace("void ", "")
            acc[short][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        per = defaultdict(float)
        for d, x in v:
            per[d] += x
        vals = list(per.values())
        print(f"   {c:32s} mean/dispatch {sum(vals)/len(vals):18.1f}  (n={len(vals)})  max/dispatch {max(vals):18.1f}")
