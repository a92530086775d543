#!/usr/bin/env python
"""Timeline of the last bench step in a rocprofv3 --kernel-trace CSV: python tools/step_timeline.py trace.csv [marker]
Prints start offset, gap to the previous kernel, duration and name of every launch between the last two launches of the
marker kernel (default: moments_kernel = the first view pass of the chain forward), and the busy / idle split."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "moments_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
prev_end, t0, busy = None, int(rows[a]["Start_Timestamp"]), 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} gap {gap:6.1f} dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:100]}")
    prev_end = e
span = int(rows[b]["Start_Timestamp"]) - t0
print(f"launches {b - a}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(span - busy) / 1e6:.3f} ms")
