#!/usr/bin/env python3
"""Trim a rocprofv3 *_kernel_stats.csv to a short table (kernel names cut at 60 chars)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
print("kernel,calls,total_ms,avg_us,percent")
for r in rows[:top]:
    name = r["Name"].split("(")[0].replace("void ", "").replace(", ", ";")[:60]
    print(f'{name},{r["Calls"]},{float(r["TotalDurationNs"]) / 1e6:.2f},{float(r["AverageNs"]) / 1e3:.1f},{float(r["Percentage"]):.2f}')
