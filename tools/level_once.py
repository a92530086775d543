#!/usr/bin/env python
"""One pyramid level of the fused bilinear path alone: python tools/level_once.py C_in C_out [steps] [warmup]
(train mode fwd + bwd at N = 2^20 x 32 views; every HIP-event timer of the step)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

C, Co = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda:0")
scene = bench.make_scene(1 << 20, 32, 32, C, 64, 128, torch.bfloat16, dev, seed=4321, workload="S1", upscale=8)
mods = bench.build_modules(C, dev, Co)
ms, kern = bench.timed_steps(scene, mods, torch.bfloat16, steps, warm, interpolate=True)
sanity = kern.pop("__sanity__")
print(json.dumps({"level": f"{C}_to_{Co}", "ms_per_step": ms, "sanity": sanity,
                  "kernels_ms": {n: round(v["ms"] / v["launches"], 3) for n, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])},
                  "timed_sum_ms": sum(v["ms"] for v in kern.values()) / steps,
                  "env": {k: v for k, v in os.environ.items() if k.startswith("DVA_")}}))
