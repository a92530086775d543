"""CPU tests of the sparse-convolution oracle and of the host logic of modules/SparseConv3d (no GPU):
the gather-GEMM-scatter restatement against torch's dense conv3d on the densified grid, kernel offsets,
coordinate downsampling, state-dict layout of the residual stages."""
import numpy as np
import torch

from oracle import sparseconv_oracle as O


def _cloud(n=300, extent=12, stride=1, seed=0):
    rng = np.random.default_rng(seed)
    xyz = np.unique(rng.integers(0, extent, size=(n, 3)), axis=0) * stride
    return torch.from_numpy(np.concatenate([xyz, np.zeros((xyz.shape[0], 1), dtype=np.int64)], 1).astype(np.int32))


def test_kernel_offsets_layout():
    o3 = O.kernel_offsets(3)
    assert o3.shape == (27, 3) and o3[0].tolist() == [-1, -1, -1] and o3[1].tolist() == [0, -1, -1]
    assert o3[13].tolist() == [0, 0, 0]
    o2 = O.kernel_offsets(2, tensor_stride=2)
    assert o2.tolist() == [[0, 0, 0], [0, 0, 2], [0, 2, 0], [0, 2, 2], [2, 0, 0], [2, 0, 2], [2, 2, 0], [2, 2, 2]]


def test_oracle_matches_dense_conv3d_stride1():
    torch.manual_seed(0)
    coords = _cloud()
    x = torch.randn(coords.shape[0], 5, dtype=torch.float64)
    W = torch.randn(27, 5, 7, dtype=torch.float64)
    nbr = O.kernel_map(coords, coords, O.kernel_offsets(3))
    out = O.sparse_conv(x, W, None, nbr)
    ref, _ = O.dense_reference(x, coords, W, 3)
    assert torch.allclose(out, ref, atol=1e-10)
    # same voxels on both sides + point-symmetric offsets: the transposed map is the map in reverse offset order
    assert torch.equal(O.kernel_map(coords, coords, -O.kernel_offsets(3)), torch.flip(nbr, [0]))


def test_oracle_matches_dense_conv3d_strided_and_transposed():
    torch.manual_seed(1)
    coords = _cloud(seed=3)
    x = torch.randn(coords.shape[0], 4, dtype=torch.float64)
    W = torch.randn(8, 4, 6, dtype=torch.float64)
    oc = O.downsample_coords(coords, 2)
    assert oc.shape[0] < coords.shape[0] and (np.asarray(oc)[:, :3] % 2 == 0).all()
    offs = O.kernel_offsets(2)
    nbr = O.kernel_map(coords, oc, offs)
    assert int((nbr >= 0).sum()) == coords.shape[0]            # every input voxel has exactly one parent slot
    out = O.sparse_conv(x, W, None, nbr)
    ref, oc_ref = O.dense_reference(x, coords, W, 2, stride=2)
    assert torch.equal(oc_ref, oc) and torch.allclose(out, ref, atol=1e-10)
    # transposed convolution back to the fine voxels: the same pairs, source and destination swapped
    Wt = torch.randn(8, 6, 3, dtype=torch.float64)
    nbr_t = O.kernel_map(oc, coords, -offs)
    up = O.sparse_conv(out, Wt, None, nbr_t)
    ref_up, _ = O.dense_reference(out, oc, Wt, 2, stride=2, tensor_stride=2, transpose=True, out_coords=coords)
    assert torch.allclose(up, ref_up, atol=1e-10)
    # the transposed map is the inverse relation of the forward map
    k, j = torch.nonzero(nbr >= 0, as_tuple=True)
    assert torch.equal(nbr_t[k, nbr[k, j].long()].long(), j)


def test_downsample_order_and_module_host_logic():
    from deepviewagg_amd.modules.SparseConv3d import nn as snn
    from deepviewagg_amd.modules.SparseConv3d import ResNetDown, ResNetUp, BottleneckBlock
    coords = _cloud(seed=5)
    coords[::3, 3] = 1                                          # two batch items
    a, b = snn.downsample_coords(coords, 4), O.downsample_coords(coords, 4)
    assert torch.equal(a, b)
    assert np.array_equal(snn.kernel_offsets(3, 2), O.kernel_offsets(3, 2))
    assert np.array_equal(snn.kernel_offsets(2, 1), O.kernel_offsets(2, 1))
    down = ResNetDown(down_conv_nn=[16, 32], N=2)
    keys = list(down.state_dict())
    assert keys[0] == "conv_in.0.kernel" and "conv_in.1.bn.running_mean" in keys
    assert "blocks.0.downsample.0.kernel" in keys and "blocks.1.block.3.kernel" in keys
    assert "blocks.1.downsample.0.kernel" not in keys
    assert down.conv_in[0].kernel.shape == (8, 16, 16) and down.blocks[0].block[0].kernel.shape == (27, 16, 32)
    up = ResNetUp(up_conv_nn=[32, 16, 24], N=1)
    assert up.conv_in[0].kernel.shape == (8, 32, 32) and up.blocks[0].block[0].kernel.shape == (27, 48, 24)
    up1 = ResNetUp(up_conv_nn=[[48, 24]], N=0, skip_first=True)
    assert up1.conv_in[0].kernel.shape == (8, 48, 24) and up1.blocks is None
    bt = BottleneckBlock(32, 64, snn.Conv3d)
    assert bt.block[0].kernel.shape == (32, 16) and bt.block[3].kernel.shape == (27, 16, 16)
    try:
        ResNetDown(down_conv_nn=[1, 2, 3])
        assert False
    except AssertionError:
        pass


def test_cpu_twin_of_a_stage_runs_on_the_oracle(monkeypatch):
    """The GPU parity test evaluates the CPU twin of a block by swapping ``ops`` for the oracle inside nn.py;
    check here that this twin works (forward + backward, encoder and decoder)."""
    from deepviewagg_amd.modules.SparseConv3d import nn as snn
    from deepviewagg_amd.modules.SparseConv3d import ResNetDown, ResNetUp
    monkeypatch.setattr(snn, "ops", O.OracleOps)
    monkeypatch.setattr(snn, "batchnorm_act_rows", O.batchnorm_act_rows)
    torch.manual_seed(0)
    coords = _cloud(seed=7)
    x = snn.SparseVoxelTensor(torch.randn(coords.shape[0], 8, requires_grad=True), coords)
    down, up = ResNetDown(down_conv_nn=[8, 16], N=1), ResNetUp(up_conv_nn=[16, 8, 12], N=1)
    y = down(x)
    assert y.s == 2 and y.C.shape[0] == O.downsample_coords(coords, 2).shape[0] and y.F.shape[1] == 16
    z = up(y, x)
    assert z.s == 1 and torch.equal(z.C, coords) and z.F.shape == (coords.shape[0], 12)
    z.F.square().mean().backward()
    assert x.F.grad is not None and down.conv_in[0].kernel.grad.abs().sum() > 0
