#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per kernel (per dispatch) for the encoder kernel."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            short = "encoder_fused" if "encoder_fused" in k else "plan_stats" if "plan_stats" in k else "plan_scan" if "plan_scan" in k else None
            if short:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kern, cs in acc.items():
    print(f"== {kern}")
    for name in sorted(cs):
        v = cs[name]
        print(f"  {name:32s} mean/dispatch {sum(v) / len(v):16.1f}  (n={len(v)})")
