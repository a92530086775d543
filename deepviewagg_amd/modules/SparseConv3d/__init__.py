"""Sparse 3D convolution blocks on voxel tensors (reference ``modules/SparseConv3d``), HIP backend."""
from .modules import BottleneckBlock, ResBlock, ResNetDown, ResNetUp  # noqa: F401
from . import nn  # noqa: F401
