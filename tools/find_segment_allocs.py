#!/usr/bin/env python
"""Which allocation makes the caching allocator call hipMalloc in steady state?  Runs the S1 step 40 times with the
allocator's history on and prints the segment_alloc events (size, step, the frames of the request that caused them)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
scene = bench.make_scene(1 << 20, 32, 32, 64, 64, 128, torch.bfloat16, dev, seed=1234)
mods = bench.build_modules(64, dev)
for _ in range(10):
    bench.step(scene, None, mods, torch.bfloat16)
torch.cuda.synchronize()
torch.cuda.memory._record_memory_history(max_entries=200000)
marks = []
for i in range(40):
    before = torch.cuda.memory_stats(dev)["num_device_alloc"]
    bench.step(scene, None, mods, torch.bfloat16)
    after = torch.cuda.memory_stats(dev)["num_device_alloc"]
    if after != before:
        marks.append(i)
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
print("steps with a device allocation:", marks)
for tr in snap["device_traces"]:
    for ev in tr:
        if ev["action"] == "segment_alloc":
            frames = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get("frames", [])
                      if "deepviewagg_amd" in f["filename"] or "bench.py" in f["filename"]][:6]
            print("segment_alloc", ev["size"] / 2**20, "MiB", frames)
