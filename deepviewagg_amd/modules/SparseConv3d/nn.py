"""Voxel-tensor layers with the surface of the reference's sparse backend shim
(``modules/SparseConv3d/nn/torchsparse.py:6-68``: ``Conv3d``, ``Conv3dTranspose``, ``BatchNorm``, ``ReLU``,
``cat``, ``SparseTensor``) over torchsparse 1.1.0, which is NOT in the reference tree.  Its published behaviour
is restated here (and in ``oracle/sparseconv_oracle.py``, "parity unpinned"):

* a tensor = features ``F`` [n, C], coordinates ``C`` int32 [n, 4] = (x, y, z, batch), tensor stride ``s``,
  plus the coordinate and kernel-map caches shared by every tensor derived from it;
* ``Conv3d(kernel_size, stride)``: output coordinates = the input's (stride 1) or the unique rows of
  ``floor(xyz / (s * stride)) * (s * stride)``; ``out[j] = sum_k F[i] @ kernel[k]`` over the pairs with
  ``C_in[i] == C_out[j] + offset_k``, ``offset_k`` = the kernel offsets scaled by ``s``; parameters
  ``kernel`` [K^3, Cin, Cout] ([Cin, Cout] for 1x1x1) and optional ``bias``;
* transposed convolution: the cached map of the matching strided convolution with source and destination
  swapped, output coordinates = the cached coordinates at stride ``s / stride``.

What is this build's own: the order of the output voxels of a strided convolution (torchsparse: ascending
64-bit coordinate hash; here ascending (batch, z, y, x)).  It is not observable through a network: every
consumer re-joins coordinates through ``coord_maps`` / ``dva_voxel_parent_index``.

The convolution itself is ``csrc/sparseconv.hip`` (output-stationary gather-MFMA, no scatter pass).
"""
import math

import numpy as np
import torch
from torch import nn

from ... import ops
from ..multimodal.pooling import batchnorm_act_rows

__all__ = ["cat", "Conv3d", "Conv3dTranspose", "ReLU", "SparseTensor", "BatchNorm"]

# modules/SparseConv3d/nn/__init__.py:28-60: the reference switches between two sparse libraries; here there
# is one backend, which carries the torchsparse field names (F, C, s, coord_maps) the multimodal blocks read.
sp3d_backend = "torchsparse"


def backend_valid(_backend):
    return _backend in {"torchsparse", "minkowski"}


def get_backend():
    return sp3d_backend


def set_backend(_backend):
    assert backend_valid(_backend)


class SparseVoxelTensor:
    """``F`` features, ``C`` int32 [n, 4] (x, y, z, batch), ``s`` tensor stride, shared caches."""

    def __init__(self, feats, coords, stride=1, coord_maps=None, kernel_maps=None):
        self.F, self.C, self.s = feats, coords, stride
        self.coord_maps = coord_maps if coord_maps is not None else {}
        self.kernel_maps = kernel_maps if kernel_maps is not None else {}
        self.coord_maps.setdefault(stride, coords)

    def like(self, feats):
        """Same voxels, new features (the caches are shared, as in torchsparse)."""
        return SparseVoxelTensor(feats, self.C, self.s, self.coord_maps, self.kernel_maps)

    def to(self, device):
        self.F, self.C = self.F.to(device), self.C.to(device)
        self.coord_maps = {k: v.to(device) for k, v in self.coord_maps.items()}
        return self

    def __add__(self, other):
        return self.like(self.F + other.F)

    def __iadd__(self, other):          # `out += self.downsample(x)` of the reference blocks
        return self.like(self.F + other.F)


def SparseTensor(feats, coordinates, batch, device=torch.device("cpu")):
    """nn/torchsparse.py:61-65."""
    if batch.dim() == 1:
        batch = batch.unsqueeze(-1)
    coords = torch.cat([coordinates.int(), batch.int()], -1)
    return SparseVoxelTensor(feats, coords).to(device)


def cat(*args, dim=1):
    """Channel concatenation of tensors on the same voxels (nn/torchsparse.py:57-58)."""
    assert dim == 1
    return args[0].like(torch.cat([a.F for a in args], dim=1))


def kernel_offsets(kernel_size, tensor_stride=1, dilation=1):
    """torchsparse 1.1.0 ``get_kernel_offsets``: per axis ``arange(-k // 2 + 1, k // 2 + 1) * stride * dilation``
    (k = 3: -1, 0, 1; k = 2: 0, 1); odd volume: x fastest, even volume: z fastest."""
    ax = [np.arange(-kernel_size // 2 + 1, kernel_size // 2 + 1) * tensor_stride * dilation for _ in range(3)]
    if kernel_size ** 3 % 2 == 1:
        offs = [[x, y, z] for z in ax[2] for y in ax[1] for x in ax[0]]
    else:
        offs = [[x, y, z] for x in ax[0] for y in ax[1] for z in ax[2]]
    return np.asarray(offs, dtype=np.int32)


def downsample_coords(coords, ratio):
    """Unique rows of ``(floor(xyz / ratio) * ratio, batch)``, ascending by (batch, z, y, x)."""
    sp = torch.div(coords[:, :3], ratio, rounding_mode="floor") * ratio
    rows = torch.cat([sp, coords[:, 3:]], 1).long()
    if rows.shape[0] == 0:
        return rows.int()
    lo = rows.min(0).values
    span = (rows.max(0).values - lo + 1)
    sx, sy, sz, sb = (int(v) for v in span.tolist())
    assert sx * sy * sz * sb < 2 ** 62, "voxel grid too large to pack into one 64-bit key"
    r = rows - lo
    key = ((r[:, 3] * sz + r[:, 2]) * sy + r[:, 1]) * sx + r[:, 0]
    uq = torch.unique(key)
    x = uq % sx
    y = (uq // sx) % sy
    z = (uq // (sx * sy)) % sz
    b = uq // (sx * sy * sz)
    return (torch.stack([x, y, z, b], 1) + lo).int().contiguous()


class Conv3d(nn.Module):
    """nn/torchsparse.py:6-18 over torchsparse ``Conv3d``; ``transpose`` as :21-40."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, bias=False,
                 transpose=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation, self.t = kernel_size, stride, dilation, transpose
        self.kernel_volume = kernel_size ** 3
        shape = (self.kernel_volume, in_channels, out_channels) if self.kernel_volume > 1 \
            else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.zeros(*shape))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        std = 1.0 / math.sqrt(self.out_channels if self.t else self.in_channels * self.kernel_volume)
        self.kernel.data.uniform_(-std, std)
        if self.bias is not None:
            self.bias.data.uniform_(-std, std)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}"
                + (", transpose=True" if self.t else ""))

    @staticmethod
    def _build(x, key, out_stride):
        """Kernel map pair of a (non-transposed) convolution with parameters ``key`` on the voxels of ``x``."""
        _, kernel_size, stride, dilation = key
        out_coords = x.C if stride == 1 else downsample_coords(x.C, out_stride)
        offs = kernel_offsets(kernel_size, x.s, dilation)
        nbr = ops.voxel_kernel_map(x.C, out_coords, offs)          # destination j <- source i
        if stride == 1 and kernel_size % 2 == 1:
            # same voxels on both sides and a point-symmetric offset list (offset_k = -offset_{K-1-k}): the
            # transposed map is the map itself with the offsets in reverse order
            nbr_t = torch.flip(nbr, [0])
        else:
            nbr_t = ops.voxel_kernel_map(out_coords, x.C, -offs)   # source i <- destination j
        x.kernel_maps[key] = (nbr, nbr_t)
        x.coord_maps.setdefault(out_stride, out_coords)

    def _maps(self, x):
        """(nbr, nbr_t, out_coords, out_stride) of this layer on tensor ``x`` (cached on the tensor)."""
        if not self.t:
            key = (x.s, self.kernel_size, self.stride, self.dilation)
            out_stride = x.s * self.stride
            if key not in x.kernel_maps:
                self._build(x, key, out_stride)
            nbr, nbr_t = x.kernel_maps[key]
            return nbr, nbr_t, x.coord_maps[out_stride], out_stride
        # transposed: the cached map of the convolution that went from stride s / stride to s, roles swapped.
        # (The decoder's residual blocks are transposed 3x3x3 stride-1 convolutions: they reuse the map of the
        #  encoder blocks at the same tensor stride; torchsparse fails when that map is absent, here it is built.)
        assert x.s % self.stride == 0, "transposed convolution below tensor stride 1"
        out_stride = x.s // self.stride
        key = (out_stride, self.kernel_size, self.stride, self.dilation)
        if key not in x.kernel_maps or out_stride not in x.coord_maps:
            if self.stride != 1:
                raise ValueError(f"transposed convolution without the matching strided convolution {key} upstream")
            self._build(x, key, out_stride)
        nbr, nbr_t = x.kernel_maps[key]
        return nbr_t, nbr, x.coord_maps[out_stride], out_stride

    def forward(self, x):
        if self.kernel_volume == 1 and self.stride == 1:
            f = x.F @ self.kernel.to(x.F.dtype)
            return x.like(f if self.bias is None else f + self.bias.to(f.dtype))
        nbr, nbr_t, out_coords, out_stride = self._maps(x)
        W = self.kernel if self.kernel.dim() == 3 else self.kernel.unsqueeze(0)   # strided 1x1x1: K = 1
        f = ops.sparse_conv(x.F, W, self.bias, nbr, nbr_t)
        return SparseVoxelTensor(f, out_coords, out_stride, x.coord_maps, x.kernel_maps)


class Conv3dTranspose(Conv3d):
    """nn/torchsparse.py:21-40 (always transposed, whatever ``transpose`` says)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, bias=False,
                 transpose=False):
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride, dilation=dilation,
                         bias=bias, transpose=True)


class _RowBatchNorm(nn.BatchNorm1d):
    """BatchNorm1d over the voxel rows, through the HIP row kernels; ``slope`` fuses the activation."""

    def forward(self, x, slope=1.0):
        if x.F.shape[0] == 0:
            return x
        return x.like(batchnorm_act_rows(x.F.contiguous(), self, slope))   # raises off-device: no CPU path


class BatchNorm(nn.Module):
    """nn/torchsparse.py:43-52 (state-dict keys ``bn.weight`` ...)."""

    def __init__(self, num_features, *, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = _RowBatchNorm(num_features=num_features, eps=eps, momentum=momentum)

    def forward(self, feats, slope=1.0):
        return self.bn(feats, slope)

    def __repr__(self):
        return self.bn.__repr__()


class ReLU(nn.ReLU):
    """nn/torchsparse.py:54-56."""

    def __init__(self, inplace=True):
        super().__init__(inplace=inplace)

    def forward(self, x):
        return x.like(torch.relu(x.F))


class Seq(nn.Sequential):
    """The reference's ``Seq`` container (core/common_modules/base_modules.py:159-167); a BatchNorm directly
    followed by a ReLU runs as one fused row kernel."""

    def __init__(self):
        super().__init__()
        self._num_modules = 0

    def append(self, module):
        self.add_module(str(self._num_modules), module)
        self._num_modules += 1
        return self

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            if isinstance(mods[i], BatchNorm) and i + 1 < len(mods) and isinstance(mods[i + 1], ReLU):
                x = mods[i](x, slope=0.0)
                i += 2
            else:
                x = mods[i](x)
                i += 1
        return x
