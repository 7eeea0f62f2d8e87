#!/usr/bin/env python3
"""Trim a rocprofv3 kernel_stats.csv (and optional counter csv) to a readable summary for profiles/."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.reader(open(src)))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        name = r[0]
        if len(name) > 140:
            name = name[:137] + "..."
        w.writerow([name] + r[1:])
print(f"wrote {dst} ({len(rows) - 1} kernels)")
