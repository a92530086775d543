#!/usr/bin/env python
"""Run one bench step with torch's sync debug mode on: every host synchronisation inside the step is reported."""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
lazy = "--materialize" not in sys.argv
N, views, C, H, W = 1 << 16, 32, 64, 64, 128
scene = bench.make_scene(N, views, 32, C, H, W, torch.bfloat16, dev, seed=1)
mods = bench.build_modules(C, dev)
bench.step(scene, None, mods, torch.bfloat16, lazy=lazy)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    bench.step(scene, None, mods, torch.bfloat16, lazy=lazy)
torch.cuda.set_sync_debug_mode("default")
print(f"{len(w)} synchronising calls in one step")
for x in w:
    print(f"  {x.filename}:{x.lineno}: {str(x.message)[:100]}")
