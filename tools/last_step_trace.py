#!/usr/bin/env python
"""Ordered kernel list of the last step in a rocprofv3 --kernel-trace CSV of tools/workload_once.py <name> <steps>:
the launches after the last but one occurrence of the step's first kernel.  Prints start-ordered name, duration, gap."""
import csv
import sys

if sys.argv[1] == "--sum":
    # per-kernel totals of a *_last_step.txt written by this script
    import collections
    import re
    tot, cnt = collections.Counter(), collections.Counter()
    for line in open(sys.argv[2]):
        m = re.match(r"\s*([\d.]+) us\s+gap\s+\S+\s+(.*)", line)
        if m:
            name = re.sub(r"\(.*", "", m.group(2)).replace("void ", "")[:70]
            tot[name] += float(m.group(1))
            cnt[name] += 1
        elif "launches" in line:
            print(line.strip())
    for name, t in tot.most_common():
        print(f"{t:9.1f} us  x{cnt[name]:<3d} {name}")
    sys.exit(0)
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the step's anchor: the gather row-index kernel runs once per step
anchor = [i for i, n in enumerate(names) if "mapping_row_index_kernel" in n]
if len(anchor) < 2:
    anchor = [0, len(rows)]
lo, hi = anchor[-2], anchor[-1]
prev_end = None
tot = 0.0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = e
    tot += (e - s) / 1e3
    print(f"{(e - s) / 1e3:9.1f} us  gap {gap:7.1f}  {r['Kernel_Name'][:120]}")
print(f"{hi - lo} launches, kernel time {tot:.1f} us, span "
      f"{(int(rows[hi - 1]['End_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3:.1f} us")
