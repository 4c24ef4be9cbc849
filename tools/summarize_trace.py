#!/usr/bin/env python3
"""Per (kernel, grid size) statistics from a rocprofv3 *kernel_trace.csv: calls, average and total duration.
Needed since the sampler runs two lanes: the aggregated *_kernel_stats.csv mixes the half-batch launches of the two concurrent lanes
(whose durations overlap each other) with the full-batch single-stream launches bench.py times for `roofline` after the timed region;
the grid size tells them apart (attention: 16 heads x 16 query blocks per sample row).
usage: summarize_trace.py <dir or file> [kernel substring ...]  ->  kernel,grid_wgs,calls,avg_us,total_ms"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
want = sys.argv[2:]
files = [root] if os.path.isfile(root) else glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gsdd::", "").replace(", ", ";")
        if want and not any(w in name for w in want):
            continue
        wgs = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        a = acc[(name[:60], wgs)]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel,grid_workgroups,calls,avg_us,total_ms")
for (name, wgs), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{name},{wgs},{n},{us / n:.1f},{us / 1e3:.2f}")
