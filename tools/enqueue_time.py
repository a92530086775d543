#!/usr/bin/env python
"""Host-side enqueue time of one bench step vs its GPU time: is the step launch-bound?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
N, views, C, H, W = 1 << int(os.environ.get("LOG2N", "20")), 32, 64, 64, 128
scene = bench.make_scene(N, views, 32, C, H, W, torch.bfloat16, dev, seed=1)
mods = bench.build_modules(C, dev)
for _ in range(2):
    bench.step(scene, None, mods, torch.bfloat16)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(5):
    t0 = time.perf_counter()
    bench.step(scene, None, mods, torch.bfloat16)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
print(f"enqueue {sum(enq) / len(enq):.2f} ms/step, total {sum(tot) / len(tot):.2f} ms/step")
