#!/usr/bin/env python
"""Per-kernel times of the fused bilinear workloads of bench.py: python tools/bilinear_kernels.py [C C_out]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
C, Co = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 64)
scene = bench.make_scene(1 << int(os.environ.get("LOG2N", "20")), 32, 32, C, 64, 128, torch.bfloat16, dev, seed=4321, workload="S1", upscale=8)
mods = bench.build_modules(C, dev, Co)
ms, kern = bench.timed_steps(scene, mods, torch.bfloat16, 3, 1, interpolate=True)
print(f"C {C} -> {Co}: {ms:.2f} ms/step")
tot = 0.0
for n, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"]):
    t = v["ms"] / 3
    tot += t
    print(f"  {n:28s} {t:7.3f} ms/step  x{v['launches'] // 3}  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.0f} GB/s")
print(f"  sum {tot:.2f}")
