#!/usr/bin/env python
"""Reduce rocprofv3 --pmc counter CSVs (one directory per pass, tools/gpu_pmc.sh) to per-kernel averages per dispatch.
Usage: pmc_summary.py <dir> [<dir> ...]   (prints one line per kernel with every counter found)"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get("FETCH_SIZE", 0))):
    parts = []
    for c, v in sorted(agg[k].items()):
        n = max(1, len(disp[k][c]))
        parts.append(f"{c}={v / n:.4g}")
    n = max(len(s) for s in disp[k].values())
    print(f"{k:60s} n={n:4d}  " + "  ".join(parts))
