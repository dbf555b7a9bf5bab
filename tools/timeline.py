#!/usr/bin/env python3
"""GPU timeline summary from a rocprofv3 --kernel-trace CSV: busy union vs span (how much of the wall time at least one
kernel was running), idle gaps, per-queue busy time, and the kernels by total time inside the steady-state window.
Usage: timeline.py kernel_trace.csv [skip_fraction=0.4] [NAME N]   (the first `skip_fraction` of the kernels = warm-up; with NAME N
also the mean duration of the LAST N launches whose name contains NAME (NAME@GRID: and whose grid has GRID threads) -- bench.py times its roofline kernel on launches it
repeats alone after the timed region)."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
        if "conv_f16x3_kernel<" in name and grid:          # one instantiation serves several layers (g_a.2: 262144 threads, g_a.4: 65536)
            name = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] + f" [grid {grid}]"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
if len(sys.argv) > 4:
    want = sys.argv[3].split("@")                     # NAME or NAME@GRID
    sel = [r for r in rows if want[0] in r[2] and (len(want) == 1 or f"[grid {want[1]}]" in r[2])][-int(sys.argv[4]):]
    if sel:
        d = [(e - s) / 1e3 for s, e, *_ in sel]
        print(f"last {len(sel)} launches of '{sys.argv[3]}': mean {sum(d) / len(d):.1f} us (min {min(d):.1f}, max {max(d):.1f})")
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = rows[int(len(rows) * skip):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
span = (t1 - t0) / 1e6
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
for s, e, *_ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(rows)}  span {span:.2f} ms  busy(union) {busy / 1e6:.2f} ms = {busy / 1e6 / span:.1%}  kernel-sum {sum(e - s for s, e, *_ in rows) / 1e6:.2f} ms")
g = sorted(gaps)
if g:
    tot = sum(g) / 1e6
    print(f"idle gaps: {len(g)} totalling {tot:.2f} ms; >5us: {sum(1 for x in g if x > 5000)} ({sum(x for x in g if x > 5000) / 1e6:.2f} ms); "
          f">20us: {sum(1 for x in g if x > 20000)} ({sum(x for x in g if x > 20000) / 1e6:.2f} ms); median {g[len(g) // 2] / 1e3:.1f} us")
perq = defaultdict(float)
for s, e, n, q, st in rows:
    perq[(q, st)] += (e - s) / 1e6
print("per queue/stream busy ms:", {k: round(v, 2) for k, v in perq.items()})
agg = defaultdict(lambda: [0, 0.0])
for s, e, n, q, st in rows:
    name = n.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:70]
    agg[name][0] += 1
    agg[name][1] += (e - s) / 1e6
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{v[1]:8.2f} ms {v[1] / span:6.1%} {v[0]:6d} x {v[1] / v[0] * 1e3:8.1f} us  {k}")

# gap contexts: which kernel boundary the big idle gaps sit at
ctx = defaultdict(lambda: [0, 0.0])
cur_e, last = rows[0][1], rows[0]
for r in rows[1:]:
    s, e = r[0], r[1]
    if s > cur_e and s - cur_e > 5000:
        a = last[2].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:38]
        b = r[2].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:38]
        ctx[(a, b)][0] += 1
        ctx[(a, b)][1] += (s - cur_e) / 1e3
    if e > cur_e:
        cur_e, last = e, r
print("largest idle-gap contexts (after -> before), count, total us:")
for k, v in sorted(ctx.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {v[1]:8.0f} us {v[0]:4d} x {v[1] / v[0]:6.1f}  {k[0]}  ->  {k[1]}")

# optional Gantt listing: TIMELINE_DUMP_MS=<ms> prints every kernel that starts within that many ms after the LAST launch of
# TIMELINE_DUMP_FROM (default: adam_kernel) in the first half of the window -- one optimisation step, stream by stream
import os
if os.environ.get("TIMELINE_DUMP_MS"):
    span_ns = float(os.environ["TIMELINE_DUMP_MS"]) * 1e6
    key = os.environ.get("TIMELINE_DUMP_FROM", "adam_kernel")
    marks = [r for r in rows if key in r[2]]
    start = marks[len(marks) // 2][0] if marks else rows[0][0]
    streams = {}
    print(f"\nkernels starting within {span_ns / 1e6:.1f} ms of a '{key}' launch (start us, duration us, stream, name):")
    for s, e, n, q, st in rows:
        if start <= s < start + span_ns:
            sid = streams.setdefault((q, st), len(streams))
            name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            print(f"{(s - start) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  s{sid}  {'    ' * sid}{name}")
