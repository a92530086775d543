"""CPU restatement of the sparse 3D convolution the backbone blocks run (TEST INFRASTRUCTURE ONLY: imported by
tests/ -- never by the product path).

Reference call sites: torch_points3d/modules/SparseConv3d/nn/torchsparse.py:6-40 (``Conv3d`` / transposed
``Conv3d`` of torchsparse), used by ResBlock / BottleneckBlock / ResNetDown / ResNetUp
(modules/SparseConv3d/modules.py:10-220).  The arithmetic lives in torchsparse **v1.1.0** (install.sh:155), a
third-party dependency that is NOT in the reference tree and cannot be installed here.  Its published algorithm
(torchsparse/nn/functional/conv.py, utils/kernel.py, nn/functional/downsample.py at that tag), restated:

* kernel offsets: per axis ``arange(-k // 2 + 1, k // 2 + 1) * tensor_stride * dilation``; odd kernel volume:
  x fastest; even volume: z fastest (``get_kernel_offsets``);
* strided convolution: output coordinates = unique rows of ``floor(xyz / (s * stride)) * (s * stride)`` with the
  batch column kept (``spdownsample``); stride 1: the input coordinates;
* ``out[j] = sum_k in[i] @ kernel[k]`` over the pairs (i, j) with ``C_in[i] == C_out[j] + offset_k``
  (``sphashquery(sphash(C_out, offsets), sphash(C_in))`` -> gather, GEMM, scatter-add), plus ``bias``;
* transposed convolution: the same pairs with source and destination swapped, on the cached coordinates.

PARITY UNPINNED for torchsparse itself: the reference holds no golden vector for it and the library cannot be
run here.  What pins this file instead: ``dense_reference`` below evaluates the same convolution as a dense
``torch.nn.functional.conv3d`` / ``conv_transpose3d`` on the densified grid (tests/test_sparseconv_oracle.py),
an independent formula.  The ORDER of the output voxels of a strided convolution is this build's own
(ascending (batch, z, y, x); torchsparse orders by its 64-bit coordinate hash) -- a row permutation no consumer
observes.
"""
import numpy as np
import torch


def kernel_offsets(kernel_size, tensor_stride=1, dilation=1):
    ax = np.arange(-kernel_size // 2 + 1, kernel_size // 2 + 1) * tensor_stride * dilation
    if kernel_size ** 3 % 2 == 1:
        offs = [[x, y, z] for z in ax for y in ax for x in ax]
    else:
        offs = [[x, y, z] for x in ax for y in ax for z in ax]
    return np.asarray(offs, dtype=np.int64)


def downsample_coords(coords, ratio):
    c = np.asarray(coords, dtype=np.int64).copy()
    c[:, :3] = np.floor_divide(c[:, :3], ratio) * ratio
    uq = np.unique(c, axis=0)                                   # ascending (x, y, z, b) lexicographic
    order = np.lexsort((uq[:, 0], uq[:, 1], uq[:, 2], uq[:, 3]))  # -> ascending (b, z, y, x)
    return torch.from_numpy(uq[order].astype(np.int32))


def kernel_map(src_coords, dst_coords, offsets):
    """int32 [K, n_dst]: row of ``src`` equal to ``dst[j] + offsets[k]`` (batch column untouched) or -1."""
    src = np.asarray(torch.as_tensor(src_coords).cpu(), dtype=np.int64)
    dst = np.asarray(torch.as_tensor(dst_coords).cpu(), dtype=np.int64)
    offs = np.asarray(torch.as_tensor(offsets).cpu(), dtype=np.int64).reshape(-1, 3)
    table = {tuple(r): i for i, r in reversed(list(enumerate(src.tolist())))}   # duplicates: smallest row wins
    nbr = np.full((offs.shape[0], dst.shape[0]), -1, dtype=np.int32)
    for k, o in enumerate(offs.tolist()):
        for j, r in enumerate(dst.tolist()):
            nbr[k, j] = table.get((r[0] + o[0], r[1] + o[1], r[2] + o[2], r[3]), -1)
    return torch.from_numpy(nbr)


def sparse_conv(x, W, bias, nbr, nbr_t=None):
    """gather -> GEMM -> scatter-add per kernel offset, plain torch (autograd gives the reference gradients)."""
    out = torch.zeros((nbr.shape[1], W.shape[2]), dtype=x.dtype)
    for k in range(nbr.shape[0]):
        dst = torch.nonzero(nbr[k] >= 0).flatten()
        if dst.numel():
            out = out.index_add(0, dst, x[nbr[k][dst].long()] @ W[k].to(x.dtype))
    return out if bias is None else out + bias.to(x.dtype)


def batchnorm_act_rows(y, bn, slope, counts=None, n=None):
    """nn.BatchNorm1d on the rows followed by leaky_relu(slope) (slope 1: none, slope 0: ReLU)."""
    assert counts is None
    z = torch.nn.functional.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                                       bn.training or not bn.track_running_stats,
                                       bn.momentum if bn.momentum is not None else 0.0, bn.eps)
    return z if slope == 1.0 else torch.nn.functional.leaky_relu(z, slope)


class OracleOps:
    """Stand-in for ``deepviewagg_amd.ops`` inside modules/SparseConv3d/nn.py when a test evaluates the CPU twin
    of a block."""
    voxel_kernel_map = staticmethod(kernel_map)
    sparse_conv = staticmethod(sparse_conv)


def dense_reference(feats, coords, kernel, kernel_size, stride=1, tensor_stride=1, transpose=False,
                    out_coords=None):
    """The same convolution on the densified grid with torch's dense operators (single batch index 0,
    non-negative coordinates that are multiples of ``tensor_stride``).  Returns (out_feats, out_coords)."""
    feats = torch.as_tensor(feats, dtype=torch.float64)
    c = np.asarray(coords, dtype=np.int64)
    assert (c[:, 3] == 0).all() and (c[:, :3] >= 0).all() and (c[:, :3] % tensor_stride == 0).all()
    K3, cin, cout = kernel.shape
    W = torch.as_tensor(kernel, dtype=torch.float64)
    g = c[:, :3] // tensor_stride
    if not transpose:
        size = g.max(0) + 1 + 2 * kernel_size
        vol = torch.zeros((1, cin, *size.tolist()), dtype=torch.float64)       # [1, C, X, Y, Z]
        vol[0, :, g[:, 0] + kernel_size, g[:, 1] + kernel_size, g[:, 2] + kernel_size] = feats.t()
        offs = kernel_offsets(kernel_size)                                      # unit offsets, kernel order
        lo = offs.min(0)
        w = torch.zeros((cout, cin, kernel_size, kernel_size, kernel_size), dtype=torch.float64)
        for k, o in enumerate(offs):
            w[:, :, o[0] - lo[0], o[1] - lo[1], o[2] - lo[2]] = W[k].t()
        dense = torch.nn.functional.conv3d(vol, w)      # dense[p] = sum_o w[o - lo] vol[p + (o - lo)]
        oc = coords if stride == 1 else downsample_coords(coords, tensor_stride * stride)
        og = np.asarray(oc, dtype=np.int64)[:, :3] // tensor_stride
        p = og + kernel_size + lo                        # p + (o - lo) = og + k + o
        return dense[0, :, p[:, 0], p[:, 1], p[:, 2]].t(), oc
    # transposed: out[fine i] = sum over (k, coarse j) with fine_i = coarse_j + offset_k of in[j] @ W[k]
    assert out_coords is not None
    s_out = tensor_stride // stride
    og = np.asarray(out_coords, dtype=np.int64)[:, :3] // s_out
    gi = c[:, :3] // s_out
    offs = kernel_offsets(kernel_size)
    out = torch.zeros((og.shape[0], cout), dtype=torch.float64)
    where = {tuple(r): j for j, r in enumerate(gi.tolist())}
    for i, r in enumerate(og.tolist()):
        for k, o in enumerate(offs.tolist()):
            j = where.get((r[0] - o[0], r[1] - o[1], r[2] - o[2]))
            if j is not None:
                out[i] += feats[j] @ W[k]
    return out, out_coords
