#!/usr/bin/env python
"""One secondary workload of bench.py, alone (A/B runs under environment switches):
python tools/workload_once.py F-L|S2|S1|bilinear_C64|bilinear_kitti|f32 [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
bf = torch.bfloat16
if name == "F-L":
    r = bench.secondary_workload("F-L", dev, bf, 20, 32, 512, steps=steps)
elif name == "S2":
    r = bench.secondary_workload("S2", dev, bf, 20, 32, 64, steps=steps)
elif name == "S1":
    r = bench.secondary_workload("S1", dev, bf, 20, 32, 64, steps=steps)
elif name == "bilinear_C64":
    r = bench.secondary_workload("bilinear", dev, bf, 20, 32, 64, steps=steps, interpolate=True)
elif name == "bilinear_kitti":
    r = bench.secondary_workload("bilinear", dev, bf, 20, 32, 128, steps=steps, interpolate=True, C_out=32)
elif name == "f32":
    r = bench.secondary_workload("f32", dev, torch.float32, 20, 32, 64, steps=steps)
elif name == "qkv":
    r = bench.secondary_workload("qkv", dev, bf, 20, 32, 64, steps=steps)
elif name == "s3dis":
    r = bench.s3dis_batch_workload(dev, steps=steps)
elif name == "s3dis_eager":
    r = bench.s3dis_batch_workload(dev, steps=steps, warmup=0, graph=False)
elif name == "nonexact":
    r = bench.nonexact_workload(dev, steps=steps)
elif name == "two_settings":
    r = bench.two_setting_bilinear_workload(dev, 20, 32, steps=steps)
elif name == "pyramid_eval":
    r = bench.kitti360_pyramid_eval(dev, 20, 32, steps=steps)
elif name == "pyramid_train":
    r = bench.kitti360_pyramid_train(dev, 20, 32, steps=steps)
else:
    raise SystemExit(f"unknown workload {name}")
r["env"] = {k: v for k, v in os.environ.items() if k.startswith("DVA_")}
print(json.dumps(r))
