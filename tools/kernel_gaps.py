#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: per kernel name the count, the average duration
and the average gap to the NEXT kernel on the device (start of the next minus end of this one).
usage: python tools/kernel_gaps.py <kernel_trace.csv> [name substring]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: [0, 0, 0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if sub in n0:
        a = agg[n0.split("(")[0][-60:]]
        a[0] += 1
        a[1] += e0 - s0
        a[2] += max(0, s1 - e0)
        a[3] = max(a[3], s1 - e0)
for n, (c, d, g, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:60s} n={c:6d} avg {d / c / 1e3:8.2f} us  gap to next avg {g / c / 1e3:6.2f} us (max {mx / 1e3:.1f})")
if rows:
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    print(f"span {span / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms ({len(rows)} kernels)")
