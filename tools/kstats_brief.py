#!/usr/bin/env python3
"""One line per retrieval kernel of a rocprofv3 kernel_stats.csv: calls and average microseconds."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(t in n for t in ("topk_", "filter_", "rescore", "select", "spmm", "linear")):
        print(f"   {n.replace('void ragraph::', '')[:100]:100s} calls {r['Calls']:>5s}  avg_us {float(r['AverageNs']) / 1e3:9.1f}")
