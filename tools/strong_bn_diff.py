#!/usr/bin/env python
"""How far `bench.py --strong` is from the single-GPU run of the same scene (VERDICT r3 "Multi-GPU"): every rank of a strong-scaling
job normalises its spatial tile with the tile's OWN BatchNorm statistics (the reference's plain nn.BatchNorm1d does the same under
data parallelism), so the pooled features of a point and the summed parameter gradients differ from the one-process result.
Runs the S1 scene once whole and once as `world` tiles on ONE device (same weights, same upstream gradient rows) and prints the
relative differences:  python tools/strong_bn_diff.py [log2_points] [world]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepviewagg_amd.parallel import tile_partition  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev, bf = torch.device("cuda:0"), torch.bfloat16
N, views, C = 1 << log2n, 32, 64
scene = bench.make_scene(N, views, 32, C, 64, 128, bf, dev, seed=1234)
g_all = torch.randn((N, 4 + C), device=dev, generator=torch.Generator(device=dev).manual_seed(99)) / N


def run(sc, g):
    mods = bench.build_modules(C, dev)            # same seed -> same weights every time
    sc = dict(sc)
    sc["grad_out"] = g
    out = bench.step(sc, None, mods, bf)
    grads = {n: p.grad.detach().float().clone() for n, p in mods[1].named_parameters() if p.grad is not None}
    return out.float(), grads, sc["x"].grad.detach().float().clone()


out_full, g_full, gx_full = run(scene, g_all)
side = int(round(N ** 0.5))
pid = torch.arange(N, device=dev)
xyz = torch.stack([(pid % side).float(), (pid // side).float(), torch.zeros(N, device=dev)], 1)
parts = tile_partition(xyz, world)
out_tiles = torch.empty_like(out_full)
g_sum, gx_sum = None, torch.zeros_like(gx_full)
for r in range(world):
    out_r, g_r, gx_r = run(bench.tile_of_scene(scene, r, world), g_all[parts[r]].contiguous())
    out_tiles[parts[r]] = out_r
    gx_sum += gx_r
    g_sum = g_r if g_sum is None else {k: g_sum[k] + v for k, v in g_r.items()}
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
res = dict(points=N, world=world,
           pooled_features_rel=rel(out_tiles[:, 4:], out_full[:, 4:]),
           feature_map_grad_rel=rel(gx_sum, gx_full),
           param_grad_rel={k: rel(g_sum[k], g_full[k]) for k in sorted(g_full)},
           note="tiles normalise with their own BatchNorm statistics; sums over the tiles against the one-process run of the same "
                "scene, same weights, same upstream gradient rows (bf16 autocast, train mode)")
res["param_grad_rel_max"] = max(res["param_grad_rel"].values())
# the yardstick: the one-process run against itself with ANOTHER summation order is bit-identical, so compare with bf16 rounding:
# the same run in a second evaluation (deterministic) -> 0; the autocast error level of the path is ~1e-2
print(json.dumps(res, indent=1))
