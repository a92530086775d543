"""Residual sparse-convolution blocks of the 3D backbone (reference ``modules/SparseConv3d/modules.py:10-220``):
same class names, constructor arguments, sub-module names (state-dict keys ``conv_in.0.kernel``,
``conv_in.1.bn.weight``, ``blocks.0.block.0.kernel``, ``blocks.0.downsample.0.kernel`` ...) and channel
bookkeeping, over ``nn.py`` (HIP sparse convolution + fused BatchNorm/ReLU row kernels)."""
import sys

import torch

from . import nn as snn
from .nn import Seq


def _stack(*layers):
    """``layers`` = (conv class, cin, cout, kernel_size, stride, extra kwargs, with_relu): conv - BN [- ReLU]
    triples appended to one ``Seq`` (indices 0, 1, 2, ... as in the reference's state dicts)."""
    seq = Seq()
    for conv, cin, cout, k, stride, kw, relu in layers:
        seq.append(conv(cin, cout, kernel_size=k, stride=stride, **kw))
        seq.append(snn.BatchNorm(cout))
        if relu:
            seq.append(snn.ReLU())
    return seq


class _Residual(torch.nn.Module):
    """``block(x) + shortcut(x)``; the shortcut is a 1x1x1 conv - BN named ``downsample`` when widths differ."""

    def _shortcut(self, conv, input_nc, output_nc, bias):
        self.downsample = None if input_nc == output_nc else \
            _stack((conv, input_nc, output_nc, 1, 1, dict(bias=bias), False))

    def forward(self, x):
        return self.block(x) + (x if self.downsample is None else self.downsample(x))


class ResBlock(_Residual):
    """Two 3x3x3 convolutions (modules.py:10-55)."""

    def __init__(self, input_nc, output_nc, convolution, bias=False):
        super().__init__()
        kw = dict(bias=bias)
        self.block = _stack((convolution, input_nc, output_nc, 3, 1, kw, True),
                            (convolution, output_nc, output_nc, 3, 1, kw, True))
        self._shortcut(snn.Conv3d, input_nc, output_nc, bias)


class BottleneckBlock(_Residual):
    """1x1x1 reduce, 3x3x3, 1x1x1 expand (modules.py:58-98)."""

    def __init__(self, input_nc, output_nc, convolution, reduction=4, bias=False):
        super().__init__()
        mid, kw = output_nc // reduction, dict(bias=bias)
        self.block = _stack((snn.Conv3d, input_nc, mid, 1, 1, kw, True),
                            (convolution, mid, mid, 3, 1, kw, True),
                            (snn.Conv3d, mid, output_nc, 1, 1, kw, True))
        self._shortcut(convolution, input_nc, output_nc, bias)


def _widths(conv_nn, stride, N, n_expected, who, what):
    """Channel bookkeeping shared by the encoder and decoder stages: the strided convolution keeps the input
    width when residual blocks follow (they do the widening), else it goes straight to the output width."""
    if isinstance(conv_nn[0], (list, tuple)) or type(conv_nn[0]).__name__ == "ListConfig":
        conv_nn = conv_nn[0]
    assert len(conv_nn) == n_expected, \
        f"{who} expects {what} but got len={len(conv_nn)}."
    nc_in, nc_out = conv_nn[0], conv_nn[-1]
    nc_skip = conv_nn[1] if n_expected == 3 else 0
    nc_strided = nc_in if stride > 1 and N > 0 else nc_out
    return nc_in, nc_strided, nc_strided + nc_skip, nc_out


class ResNetDown(torch.nn.Module):
    """Encoder stage: strided conv - BN - ReLU, then N residual blocks (modules.py:103-170)."""

    CONVOLUTION = "Conv3d"

    def __init__(self, down_conv_nn=[], kernel_size=2, dilation=1, stride=2, N=1, bias=False, block="ResBlock",
                 **kwargs):
        super().__init__()
        conv = getattr(snn, self.CONVOLUTION)
        nc_in, nc_strided, nc_block, nc_out = self._parse_conv_nn(down_conv_nn, stride, N)
        self.conv_in = _stack((conv, nc_in, nc_strided, kernel_size, stride, dict(bias=bias, dilation=dilation), True))
        self.blocks = None
        if N > 0:
            block_cls = getattr(sys.modules[__name__], block)
            self.blocks = Seq()
            for i in range(N):
                self.blocks.append(block_cls(nc_block if i == 0 else nc_out, nc_out, conv, bias=bias))

    def _parse_conv_nn(self, down_conv_nn, stride, N):
        return _widths(down_conv_nn, stride, N, 2, "ResNetDown", "down_conv_nn = (nc_in, nc_out)")

    def forward(self, x):
        x = self.conv_in(x)
        return x if self.blocks is None else self.blocks(x)


class ResNetUp(ResNetDown):
    """Decoder stage: transposed strided conv with the skip features concatenated before (``skip_first``) or
    after it (modules.py:173-229)."""

    CONVOLUTION = "Conv3dTranspose"

    def __init__(self, up_conv_nn=[], kernel_size=2, dilation=1, stride=2, N=1, bias=False, skip_first=False,
                 **kwargs):
        self.skip_first = skip_first
        super().__init__(down_conv_nn=up_conv_nn, kernel_size=kernel_size, dilation=dilation, stride=stride, N=N,
                         bias=bias, **kwargs)

    def _parse_conv_nn(self, up_conv_nn, stride, N):
        if self.skip_first:
            return _widths(up_conv_nn, stride, N, 2, "ResNetUp(skip_first=True)", "up_conv_nn = (nc_in, nc_out)")
        return _widths(up_conv_nn, stride, N, 3, "ResNetUp(skip_first=False)",
                       "up_conv_nn = (nc_in, nc_skip_in, nc_out)")

    def forward(self, x, skip):
        if skip is not None and self.skip_first:
            x = snn.cat(x, skip)
        x = self.conv_in(x)
        if skip is not None and not self.skip_first:
            x = snn.cat(x, skip)
        return x if self.blocks is None else self.blocks(x)
