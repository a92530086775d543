#!/usr/bin/env python
"""Eval-mode (no_grad) forward of the bilinear pooling at C_out = 128 (KITTI-360 level 256 -> 128): the one fused kernel
against the materialised fallback.  python tools/bilinear_eval_c128.py [log2_points [C_in C_out]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepviewagg_amd import fused_chain, ops  # noqa: E402

dev = torch.device("cuda", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 19
C, Co = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 128)
scene = bench.make_scene(1 << L, 32, 32, C, 64, 128, torch.bfloat16, dev, seed=4321, workload="S1", upscale=8)
mods = bench.build_modules(C, dev, Co)
for m in mods:
    m.eval()


def forward():
    atomic_pool, view_pool, fusion = mods
    x = scene["x"]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        packed = ops.pack_gather_index(scene["images"], scene["atom_ptr"], scene["pixels"], ratio=1.0)
        res = torch.tensor([scene["mapping_size"]], dtype=torch.float32, device=x.device)
        coords = (scene["pixels"] / (res - 1))[:, [1, 0]]
        x_mod = ops.lazy_gather_bilinear(x, packed, coords, True)
        x_mod = atomic_pool(None, x_mod, None, scene["atom_ptr"])
        x_pool = view_pool(scene["x_3d"], x_mod, scene["x_map"], scene["csr"])
        return fusion(scene["x_3d"], x_pool)


outs = {}
for fused in (True, False):
    fused_chain.FORCE = None if fused else False
    with torch.no_grad():
        for _ in range(2):
            outs[fused] = forward()
        torch.cuda.synchronize()
        ops.TIMER = ops.KernelTimer()
        t0 = time.perf_counter()
        for _ in range(3):
            forward()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        timer, ops.TIMER = ops.TIMER, None
    top = sorted(timer.summary().items(), key=lambda kv: -kv[1]["ms"])[:4]
    print(f"N = 2^{L}, {C} -> {Co}, eval forward, fused={fused}: {ms:.2f} ms", {n: round(v["ms"] / 3, 2) for n, v in top})
fused_chain.FORCE = None
d = (outs[True].float() - outs[False].float()).norm() / outs[False].float().norm()
print(f"rel |fused - fallback| = {float(d):.2e}")
