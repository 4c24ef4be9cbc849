#!/usr/bin/env python3
"""Mean counter value per dispatch and kernel from rocprofv3 --pmc output (every *counter_collection.csv under a directory).
usage: summarize_pmc.py <dir> [kernel substring ...]   ->  kernel,counter,mean_per_dispatch,dispatches"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
want = sys.argv[2:]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gsdd::", "").replace(", ", ";")
        if want and not any(w in name for w in want):
            continue
        per_dispatch[(name, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (name, ctr, _), v in per_dispatch.items():
        a = acc[(name, ctr)]
        a[0] += v
        a[1] += 1
print("kernel,counter,mean_per_dispatch,dispatches")
for (name, ctr), (tot, n) in sorted(acc.items()):
    print(f"{name[:70]},{ctr},{tot / n:.1f},{n}")
