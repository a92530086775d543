"""deepviewagg_amd — MI355X (gfx950) native implementation of the DeepViewAgg multimodal hot path.

Scope: point->pixel mapping build, per-point multi-view feature gather, DeepViewAgg
view-attention pooling, forward and backward, behind the reference's ``torch_points3d`` plugin
surface.  Compute goes through the C-ABI library ``csrc/libdva_hip.so`` (``include/dva.h``).
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def library_path():
    return _lib.LIB_PATH
