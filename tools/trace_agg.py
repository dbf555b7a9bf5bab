#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid, workgroup): calls, total ms, mean us.  Usage: trace_agg.py trace.csv [top]"""
import csv
import sys
from collections import defaultdict

agg = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("LDS_Block_Size", ""))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[key][0] += 1
        agg[key][1] += d
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(v[1] for v in agg.values())
print(f"total {tot / 1e3:.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1] / 1e3:9.2f} ms {v[0]:6d} x {v[1] / v[0]:9.1f} us  grid {k[1]:>8},{k[2]},{k[3]} lds {k[4]:>6}  {k[0]}")
