#!/usr/bin/env python
"""Per-kernel SQ counter table from a rocprofv3 --pmc counter_collection CSV (largest launch per kernel).
Usage: python tools/pmc_sq.py <counter_collection.csv>"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "dva::" not in name:
        continue
    short = name.split("dva::")[1].split("(")[0]
    key = (short, r["Dispatch_Id"])
    rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
best = {}
for (short, _), c in rows.items():
    if short not in best or c.get("SQ_WAVE_CYCLES", 0) > best[short].get("SQ_WAVE_CYCLES", 0):
        best[short] = c
names = sorted({k for c in best.values() for k in c})
print("kernel".ljust(52) + "".join(n.replace("SQ_", "")[:14].rjust(15) for n in names))
for short, c in sorted(best.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    print(short[:50].ljust(52) + "".join(
        (f"{c.get(n, 0) / wc:14.3f}" if n != "SQ_WAVE_CYCLES" and n.startswith("SQ_") and "BUSY_CYCLES" not in n and "WAVES" not in n
         else f"{c.get(n, 0):14.3g}").rjust(15) for n in names))
