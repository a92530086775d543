#!/usr/bin/env python
"""Sparse 3D convolution on MI355X: forward + backward of one 3x3x3 convolution and of a ResNetDown stage over a
synthetic surface cloud (voxels on the faces of a room-sized box).  Prints per-kernel HIP-event times with the
dense-equivalent FLOP rate (2 * pairs * Cin * Cout) and the gather traffic.
Usage: python tools/sparse_conv_bench.py [n_voxels] [channels] [bf16|fp32]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepviewagg_amd import ops  # noqa: E402
from deepviewagg_amd.modules.SparseConv3d import ResNetDown, nn as snn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
C = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else torch.bfloat16
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
extent = int(np.sqrt(n / 4.5))
p = rng.integers(0, extent, size=(int(n * 1.6), 3))
face = rng.integers(0, 3, p.shape[0])
p[np.arange(p.shape[0]), face] = rng.integers(0, 2, p.shape[0]) * (extent - 1)
c = np.unique(p, axis=0)[:n]
coords = torch.from_numpy(np.concatenate([c, np.zeros((c.shape[0], 1), dtype=np.int64)], 1).astype(np.int32)).to(dev)
# spatially coherent order (as a voxelised scan arrives): sort by a coarse block key
key = ((coords[:, 2] // 8).long() * 4096 + (coords[:, 1] // 8).long()) * 4096 + (coords[:, 0] // 8).long()
coords = coords[torch.argsort(key)].contiguous()
n = coords.shape[0]
offs = snn.kernel_offsets(3, 1)
ops.voxel_kernel_map(coords[:1000], coords[:1000], offs)      # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
nbr = ops.voxel_kernel_map(coords, coords, offs)
torch.cuda.synchronize()
t_map = (time.perf_counter() - t0) * 1e3
nbr_t = torch.flip(nbr, [0])
pairs = int((nbr >= 0).sum())
print(f"{n} voxels, {pairs / n:.1f} neighbours / voxel of 27, kernel map 3x3x3 {t_map:.2f} ms")

x = torch.randn(n, C, device=dev, dtype=dtype, requires_grad=True)
W = (torch.randn(27, C, C, device=dev) / 40).requires_grad_(True)
g = torch.randn(n, C, device=dev, dtype=dtype)
for _ in range(2):
    out = ops.sparse_conv(x, W, None, nbr, nbr_t)
    out.backward(g)
ops.TIMER = ops.KernelTimer()
reps = 5
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    x.grad = W.grad = None
    out = ops.sparse_conv(x, W, None, nbr, nbr_t)
    out.backward(g)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
timer, ops.TIMER = ops.TIMER, None
flops = 2.0 * pairs * C * C
print(f"conv 3x3x3 {C}->{C} {str(dtype).split('.')[-1]}: fwd+bwd {wall:.2f} ms")
for name, a in sorted(timer.summary().items()):
    ms = a["ms"] / a["launches"]
    print(f"  {name:22s} {ms:7.3f} ms x{a['launches']:<3d} {flops / ms / 1e9:8.1f} TFLOP/s (pairs only) "
          f"{a['bytes'] / a['launches'] / ms / 1e6:8.1f} GB/s (compulsory bytes)")

stage = ResNetDown(down_conv_nn=[C, 2 * C], N=2).to(dev)
xs = snn.SparseVoxelTensor(torch.randn(n, C, device=dev, requires_grad=True), coords)
with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
    for _ in range(2):
        y = stage(xs)
        y.F.float().square().mean().backward()
    torch.cuda.synchronize()
    ops.TIMER = ops.KernelTimer()
    ops.SPARSE_CONV_TIMER_SHAPES = True
    t0 = time.perf_counter()
    for _ in range(reps):
        y = stage(xs)
        y.F.float().square().mean().backward()
    torch.cuda.synchronize()
timer, ops.TIMER = ops.TIMER, None
for name, a in sorted(timer.summary().items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {name:22s} {a['ms'] / reps:7.3f} ms/step in {a['launches'] // reps} launches")
print(f"ResNetDown({C}->{2 * C}, N=2) on {n} -> {y.F.shape[0]} voxels: fwd+bwd "
      f"{(time.perf_counter() - t0) / reps * 1e3:.2f} ms (kernel maps cached)")
