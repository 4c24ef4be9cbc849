#!/usr/bin/env python3
"""profiles/rN_pmc_sq_counters.csv (round 4 layout) from rocprofv3 --pmc passes over bench.py itself.
usage: make_sq_csv.py <regime>=<dir> [<regime>=<dir> ...] [-- kernel substring ...]  > profiles/rN_pmc_sq_counters.csv

One row per (regime, kernel, grid size): the mean per dispatch of every SQ counter of the pass and the two occupancy figures bench.py
reports (`roofline.counters`, `extra.roofline_families[*].counters`):
  mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs  over  SQ_BUSY_CYCLES / 32 shader engines (= the dispatch's length in cycles)
  valu_issue_busy = 4 * SQ_ACTIVE_INST_VALU (quad-cycles) / 1024 SIMDs  over the same
(units: MI355X_MICROARCH.md, row `s_memtime tick vs SQ PMC units`).  A regime is a weight scale of bench.py: `flat` = the reference
init (near-uniform softmax rows), `trained_like` = bench.py --trained-like."""
import collections
import csv
import glob
import os
import sys

COUNTERS = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
SIMDS, SHADER_ENGINES = 1024, 32


def collect(root, want):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    files = sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                          # the newest pass only
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gsdd::", "").replace(", ", ";")
            if want and not any(w in name for w in want):
                continue
            wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            per_dispatch[(name, wgs, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (name, wgs, ctr, _), v in per_dispatch.items():
            a = acc[(name, wgs)][ctr]
            a[0] += v
            a[1] += 1
    return acc


def main():
    args = sys.argv[1:]
    want = []
    if "--" in args:
        i = args.index("--")
        args, want = args[:i], args[i + 1:]
    print("regime,kernel,grid_workgroups,dispatches," + ",".join(COUNTERS) + ",mfma_busy,valu_issue_busy")
    for spec in args:
        regime, root = spec.split("=", 1)
        acc = collect(root, want)
        for (name, wgs), ctrs in sorted(acc.items()):
            if not all(c in ctrs for c in COUNTERS):
                continue
            mean = {c: ctrs[c][0] / ctrs[c][1] for c in COUNTERS}
            cycles = mean["SQ_BUSY_CYCLES"] / SHADER_ENGINES
            if cycles <= 0:
                continue
            mfma = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / cycles
            valu = 4.0 * mean["SQ_ACTIVE_INST_VALU"] / SIMDS / cycles
            print(f"{regime},{name[:70]},{wgs},{ctrs['SQ_BUSY_CYCLES'][1]}," + ",".join(f"{mean[c]:.1f}" for c in COUNTERS) +
                  f",{mfma:.4f},{valu:.4f}")


if __name__ == "__main__":
    main()
