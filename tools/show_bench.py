#!/usr/bin/env python
"""Print the per-kernel table of a bench.py JSON line: python tools/show_bench.py <file>."""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"{d['ms_per_step']:.2f} ms/step   {d['value'] / 1e6:.2f} M {d['unit']}   roofline {d['roofline']['kernel']} "
      f"frac={d['roofline']['frac']:.3f}")
for k, v in d["kernels"].items():
    print(f"{k:34s} {v['avg_ms']:8.3f} ms x{v['launches']:<3d} {v['GBps']:8.0f} GB/s")
