#!/usr/bin/env python
"""Fused view kernel (dva_chain_attn_fwd, eval mode = ONE launch per forward) and the attention backward against the number of
views per point -- VERDICT r3 item 7 asked for the roofline fraction per tile shape: every point k views (k = 1 .. 32: tiles of
32 / k whole points, k = 32: one point per tile), and the two ragged mixes of bench.py (S2; S2 without unseen points).
Same V for every row (2^22 views), C = 64 bf16, G = 4.  python tools/fwd_shape_sweep.py > profiles/r04_fwd_shape_sweep.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepviewagg_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
C, H, W, B = 64, 64, 128, 32
V_TARGET = 1 << 22


def scene_fixed_k(k, seed=5):
    g = torch.Generator(device=dev).manual_seed(seed)
    N = V_TARGET // k
    V = N * k
    csr = torch.arange(0, V + 1, k, dtype=torch.int64, device=dev)
    images = torch.stack([torch.randperm(B, generator=g, device=dev)[:k].sort()[0] for _ in range(256)]) \
        .repeat((N + 255) // 256, 1)[:N].reshape(-1)
    pixels = torch.stack([torch.randint(0, W, (V,), generator=g, device=dev),
                          torch.randint(0, H, (V,), generator=g, device=dev)], 1).to(torch.int16)
    return dict(csr=csr, images=images, pixels=pixels, atom_ptr=torch.arange(V + 1, dtype=torch.int64, device=dev),
                x=torch.randn(B, C, H, W, generator=g, device=dev).to(bf).contiguous(memory_format=torch.channels_last),
                x_map=torch.rand(V, 8, generator=g, device=dev), x_3d=torch.randn(N, 4, generator=g, device=dev),
                mapping_size=(W, H))


def measure(scene, label):
    mods = bench.build_modules(C, dev)
    N, V = scene["x_3d"].shape[0], scene["x_map"].shape[0]
    row = {"shape": label, "points": N, "views": V}
    for mode in ("eval", "train"):
        mods[1].train(mode == "train")
        for _ in range(2):
            bench.step(scene, None, mods, bf)
        torch.cuda.synchronize()
        ops.TIMER = ops.KernelTimer()
        for _ in range(5):
            bench.step(scene, None, mods, bf)
        timer, ops.TIMER = ops.TIMER, None
        kern = timer.summary()
        for key, nbytes in (("chain_attn_fwd", bench.fused_fwd_bytes(V, N, C, 2)), ("chain_attn_bwd", bench.fused_bwd_bytes(V, N, C, 2))):
            if key in kern:
                t = kern[key]["ms"] / kern[key]["launches"]
                row[f"{key}_{mode}"] = {"ms": round(t, 4), "frac_of_hbm_peak": round(nbytes / (t * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 3),
                                        "algorithmic_MB": round(nbytes / 1e6, 1)}
    del mods
    torch.cuda.empty_cache()
    return row


rows = [measure(scene_fixed_k(k), f"every point {k} view{'s' if k > 1 else ''}") for k in (1, 2, 4, 8, 16, 32)]
s2 = bench.make_scene(1 << 20, 32, 32, C, H, W, bf, dev, seed=4321, workload="S2")
rows.append(measure(s2, "S2: k = min(32, 1 + Geom(0.2)), 10 % unseen"))
print(json.dumps({"what": __doc__.split("python tools")[0].strip(), "rows": rows}, indent=1))
