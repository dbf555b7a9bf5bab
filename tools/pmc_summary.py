#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per launch.
Usage: pmc_summary.py out.csv dir1 [dir2 ...]   (each dir = one rocprofv3 -d output with *counter_collection.csv)"""
import csv
import glob
import os
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
            key = (name, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
            c = acc[key][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
names = sorted({c for k in acc for c in acc[k]})
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "workgroup", "launches"] + names)
    for k in sorted(acc):
        n = max(v[0] for v in acc[k].values())
        w.writerow([k[0], k[1], k[2], n] + [f"{acc[k][c][1] / acc[k][c][0]:.1f}" if c in acc[k] else "" for c in names])
print(open(out).read())
