#!/usr/bin/env python
"""Mapping build at 1 M candidate points per image (S3DIS settings): wall time per image; run under
rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepviewagg_amd.core.multimodal.visibility import SplattingVisibility  # noqa: E402

n = int(os.environ.get("N_POINTS", 1 << 20))
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
face = rng.integers(0, 6, n)
uvw = rng.random((n, 3))
uvw[np.arange(n), face // 2] = face % 2
xyz = torch.from_numpy((uvw * np.array([8.0, 6.0, 3.0])).astype(np.float32)).to(dev)
cams = torch.tensor([[3.1, 2.2, 1.4], [5.0, 3.0, 1.2], [2.0, 4.5, 1.6], [6.5, 1.5, 1.5]], device=dev)
model = SplattingVisibility(camera="s3dis_equirectangular", img_size=(2048, 1024), r_max=8.0, r_min=0.05, voxel=0.02,
                            k_swell=1.0, d_swell=1000, exact=True)
opk = torch.zeros(3, device=dev)
for c in cams:
    out = model(xyz, c, img_opk=opk)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    for c in cams:
        out = model(xyz, c, img_opk=opk)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (reps * len(cams))
print(f"n={n} candidates: {dt * 1e3:.3f} ms/image, {n / dt / 1e9:.2f} G candidates/s, mapped {out['idx'].shape[0]}")
