#!/usr/bin/env python
"""Timing of the generic (non-fused) C-ABI ops at the headline size: catches pathologically slow kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepviewagg_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
N, VIEWS, C, G = 1 << 20, 32, 64, 4
V = N * VIEWS
g = torch.Generator(device=dev).manual_seed(0)
csr = torch.arange(0, V + 1, VIEWS, device=dev)
x = torch.randn(V, C, generator=g, device=dev).bfloat16().requires_grad_()
compat = torch.randn(V, G, generator=g, device=dev).requires_grad_()
w = torch.randn(N, C, generator=g, device=dev).bfloat16()


def timeit(name, fn, nbytes, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:34s} {dt * 1e3:8.2f} ms  {nbytes / dt / 1e9:8.0f} GB/s")


for mode in ("max", "sum", "mean"):
    timeit(f"segment_csr fwd {mode}", lambda: ops.segment_csr(x.detach(), csr, reduce=mode), V * C * 2)


def seg_bwd():
    out = ops.segment_csr(x, csr, reduce="max")
    out.backward(w)
    x.grad = None


timeit("segment_csr max fwd+bwd", seg_bwd, V * C * 2 * 2)
timeit("segment_softmax_csr fwd", lambda: ops.segment_softmax_csr(compat.detach(), csr), V * G * 4 * 2)


def att():
    out, _, _ = ops.view_attention(x, compat, csr)
    out.backward(w)
    x.grad = None
    compat.grad = None


timeit("view_attention fwd+bwd (dense)", att, V * C * 2 * 3)
B, H, W = 32, 64, 128
fm = torch.randn(B, C, H, W, generator=g, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
images = torch.arange(B, device=dev).repeat(N)
pixels = torch.stack([torch.randint(0, W, (V,), generator=g, device=dev),
                      torch.randint(0, H, (V,), generator=g, device=dev)], 1).to(torch.int16)
packed = ops.pack_gather_index(images, torch.arange(V + 1, device=dev), pixels, ratio=1.0)
timeit("gather_nearest fwd", lambda: ops.gather_nearest(fm.detach(), packed), V * (2 * C * 2 + 8))


def gn_bwd():
    out = ops.gather_nearest(fm, packed)
    out.backward(x.detach())
    fm.grad = None


timeit("gather_nearest fwd+bwd", gn_bwd, V * (3 * C * 2 + 16))
coords = torch.rand(V, 2, generator=g, device=dev)
timeit("gather_bilinear fwd", lambda: ops.gather_bilinear(fm.detach(), packed, coords), V * (5 * C * 2 + 16))


def gb_bwd():
    out = ops.gather_bilinear(fm, packed, coords)
    out.backward(x.detach())
    fm.grad = None


timeit("gather_bilinear fwd+bwd", gb_bwd, V * (6 * C * 2 + 16) + V * 4 * C * 8)
