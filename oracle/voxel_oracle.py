"""CPU restatement of the voxel parent index (TEST INFRASTRUCTURE ONLY: imported by tests/ and smoke).

Reference: torch_points3d/modules/multimodal/modules.py:176-198 -- the spatial columns of the input
coordinates are floored to the output stride (``((c.float() / s).floor() * s).int()``) and looked up among the
output coordinates with torchsparse's ``sphashquery(sphash(in), sphash(out))`` (torchsparse v1.1.0,
install.sh:155 -- a third-party dependency that is NOT in the reference tree and not installed here).
Its published contract: the index of the row of ``out`` equal to the query row, ``-1`` if there is none.

PARITY UNPINNED for the hash query itself: no golden vector of torchsparse exists in the reference and the
library cannot be run here.  The flooring expression is the reference's own (executed below with torch,
exactly as written); the lookup is restated with a Python dict.  The downstream consumer
(``select_points(idx, mode='merge')``) is pinned by golden fixtures (tests/golden/mapping_*.npz).
"""
import numpy as np
import torch


def floor_coords(in_coords, stride_out, batch_col=3):
    """modules.py:192-194, verbatim arithmetic (float32 division, floor, multiply, int cast)."""
    c = torch.as_tensor(in_coords).clone().int()
    cols = [k for k in range(4) if k != batch_col]
    c[:, cols] = ((c[:, cols].float() / stride_out).floor() * stride_out).int()
    return c


def voxel_parent_index(in_coords, out_coords, stride_out, batch_col=3):
    fl = floor_coords(in_coords, stride_out, batch_col).numpy()
    table = {}
    for j, row in enumerate(np.asarray(out_coords).tolist()):
        table.setdefault(tuple(row), j)
    return np.array([table.get(tuple(r), -1) for r in fl.tolist()], dtype=np.int64)
