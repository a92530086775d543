"""Voxel parent index (C ABI dva_voxel_parent_index) and MultimodalBlockDown's re-indexing of the mappings
after a strided 3D block (reference modules/multimodal/modules.py:101-236).  The hash query replaces
torchsparse's sphashquery, which is not in the reference tree (oracle/voxel_oracle.py: parity unpinned for
the query itself; exact integer contract + properties here; the merge it feeds is pinned by golden vectors)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, t
from oracle import voxel_oracle as VO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def coarsen(coords, stride, batch_col=3):
    """Unique floored coordinates = the voxel set a stride-`stride` sparse convolution produces."""
    fl = VO.floor_coords(coords, stride, batch_col)
    return torch.unique(fl, dim=0)


@pytest.mark.parametrize("n,stride,span", [(0, 2, 8), (1, 2, 8), (5000, 2, 40), (20000, 4, 300), (3000, 3, 50)])
def test_voxel_parent_index_matches_oracle(n, stride, span):
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(n + stride)
    xyz = torch.randint(-span, span, (n, 3), generator=gen, dtype=torch.int32)
    batch = torch.randint(0, 3, (n, 1), generator=gen, dtype=torch.int32)
    coords = torch.unique(torch.cat([xyz, batch], 1), dim=0)
    out = coarsen(coords, stride)
    out = out[torch.randperm(out.shape[0], generator=gen)]            # any order of the output voxels
    if out.shape[0] > 10:
        out = out[:-3]                                                # some parents missing -> -1
    idx = ops.voxel_parent_index(coords.to(DEV), out.to(DEV), stride)
    ref = VO.voxel_parent_index(coords.numpy(), out.numpy(), stride)
    assert idx.dtype == torch.int64 and np.array_equal(idx.cpu().numpy(), ref)
    found = idx.cpu() >= 0
    assert torch.equal(out[idx.cpu()[found]], VO.floor_coords(coords, stride)[found])
    if out.shape[0] > 10:
        assert int((~found).sum()) > 0


def test_voxel_parent_index_large_property():
    """1M voxels: every input voxel finds the output voxel holding its floored coordinates."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(0)
    coords = torch.unique(torch.cat([torch.randint(-2000, 2000, (1 << 20, 3), generator=gen, dtype=torch.int32),
                                     torch.randint(0, 4, (1 << 20, 1), generator=gen, dtype=torch.int32)], 1), dim=0)
    out = coarsen(coords, 2).to(DEV)
    idx = ops.voxel_parent_index(coords.to(DEV), out, 2)
    assert int(idx.min()) >= 0
    assert torch.equal(out[idx].cpu(), VO.floor_coords(coords, 2))
    # idempotence: the output voxels are their own parents at the output stride
    assert torch.equal(ops.voxel_parent_index(out, out, 2).cpu(), torch.arange(out.shape[0]))


def test_voxel_parent_index_rejects_cpu_and_bad_shapes():
    from deepviewagg_amd import ops, _lib
    c = torch.zeros(4, 4, dtype=torch.int32)
    with pytest.raises(_lib.DvaError):
        ops.voxel_parent_index(c, c, 2)
    with pytest.raises(AssertionError):
        ops.voxel_parent_index(torch.zeros(4, 3, dtype=torch.int32, device=DEV), c.to(DEV), 2)


class _StridedBlock(torch.nn.Module):
    """Test double of a stride-2 sparse convolution: parents = unique floored coordinates, features = mean."""

    def forward(self, x):
        from deepviewagg_amd.modules.multimodal.modules import SparseVoxels
        s_out = x.s * 2
        fl = VO.floor_coords(x.C.cpu(), s_out).to(x.C.device)
        out_c, inv = torch.unique(fl, dim=0, return_inverse=True)
        F = torch.zeros(out_c.shape[0], x.F.shape[1], device=x.F.device).index_add_(0, inv, x.F)
        maps = dict(x.coord_maps)
        maps[s_out] = out_c
        return SparseVoxels(F, out_c, s_out, maps)


def test_multimodal_block_down_merges_mappings_onto_parent_voxels():
    from deepviewagg_amd.core.multimodal.image import ImageMapping
    from deepviewagg_amd.modules.multimodal.modules import MultimodalBlockDown, SparseVoxels, IdentityBranch
    g = load_golden("mapping_build")
    n_pts = len(g["pointers"]) - 1
    m = ImageMapping.from_dense(t(g["dense_point_ids"], DEV), t(g["dense_image_ids"], DEV),
                                t(g["dense_pixels"], DEV), t(g["dense_features"], DEV), num_points=n_pts)
    gen = torch.Generator().manual_seed(1)
    coords = torch.cat([torch.randperm(4 * n_pts, generator=gen)[:n_pts].view(-1, 1).int() % 37,
                        torch.randint(0, 9, (n_pts, 2), generator=gen, dtype=torch.int32),
                        torch.zeros(n_pts, 1, dtype=torch.int32)], 1)
    coords = torch.unique(coords, dim=0)
    if coords.shape[0] < n_pts:     # pad with distinct far voxels so that there is one voxel per point
        extra = torch.arange(n_pts - coords.shape[0], dtype=torch.int32).view(-1, 1) + 1000
        coords = torch.cat([coords, torch.cat([extra, extra, extra, torch.zeros_like(extra)], 1)])
    x_seen = (m.pointers[1:] > m.pointers[:-1])

    class _Mod:   # the slice of the ImageData interface the block uses
        def __init__(self, mapping):
            self.mapping = mapping

        def select_points(self, idx, mode='pick'):
            return _Mod(self.mapping.select_points(idx, mode=mode))

    d = dict(x_3d=SparseVoxels(torch.randn(n_pts, 4, device=DEV), coords.to(DEV), 1), x_seen=x_seen,
             modalities=dict(image=_Mod(m)))
    block = MultimodalBlockDown(_StridedBlock(), None, image=IdentityBranch())
    out = block(d)
    # expected: oracle parent index -> the (golden-pinned) merge
    out_c = out['x_3d'].C.cpu()
    ref_idx = torch.from_numpy(VO.voxel_parent_index(coords.numpy(), out_c.numpy(), 2))
    assert int(ref_idx.min()) >= 0
    exp = m.select_points(ref_idx.to(DEV), mode='merge')
    got = out['modalities']['image'].mapping
    assert torch.equal(got.pointers, exp.pointers) and torch.equal(got.images, exp.images)
    assert torch.equal(got.values[1].pointers, exp.values[1].pointers)
    assert torch.equal(got.pixels, exp.pixels)
    assert torch.allclose(got.features, exp.features)
    exp_seen = torch.zeros(out_c.shape[0], dtype=torch.int64).index_add_(0, ref_idx, x_seen.cpu().long()) > 0
    assert torch.equal(out['x_seen'].cpu().bool(), exp_seen)
    assert out['x_3d'].s == 2 and got.num_groups == out_c.shape[0]


def test_multimodal_block_down_with_the_hip_resnet_stage():
    """The same contract with the real strided block: ResNetDown (HIP sparse convolution) between the
    multimodal branches; its output voxels / stride drive the re-indexing of the mappings."""
    from deepviewagg_amd.core.multimodal.image import ImageMapping
    from deepviewagg_amd.modules.multimodal.modules import MultimodalBlockDown, IdentityBranch
    from deepviewagg_amd.modules.SparseConv3d import ResNetDown, nn as snn
    g = load_golden("mapping_build")
    n_pts = len(g["pointers"]) - 1
    m = ImageMapping.from_dense(t(g["dense_point_ids"], DEV), t(g["dense_image_ids"], DEV),
                                t(g["dense_pixels"], DEV), t(g["dense_features"], DEV), num_points=n_pts)
    gen = torch.Generator().manual_seed(2)
    side = int(np.ceil(n_pts ** (1 / 3))) + 1
    lin = torch.randperm(side ** 3, generator=gen)[:n_pts]
    coords = torch.stack([lin % side, (lin // side) % side, lin // (side * side), torch.zeros_like(lin)], 1).int()
    x_seen = (m.pointers[1:] > m.pointers[:-1])

    class _Mod:
        def __init__(self, mapping):
            self.mapping = mapping

        def select_points(self, idx, mode='pick'):
            return _Mod(self.mapping.select_points(idx, mode=mode))

    torch.manual_seed(0)
    stage = ResNetDown(down_conv_nn=[4, 16], N=1).to(DEV)
    x = snn.SparseVoxelTensor(torch.randn(n_pts, 4, device=DEV), coords.to(DEV))
    out = MultimodalBlockDown(stage, None, image=IdentityBranch())(
        dict(x_3d=x, x_seen=x_seen, modalities=dict(image=_Mod(m))))
    out_c = out['x_3d'].C.cpu()
    assert out['x_3d'].s == 2 and out['x_3d'].F.shape == (out_c.shape[0], 16)
    assert torch.equal(torch.unique(out_c, dim=0), torch.unique(VO.floor_coords(coords, 2), dim=0))
    ref_idx = torch.from_numpy(VO.voxel_parent_index(coords.numpy(), out_c.numpy(), 2))
    assert int(ref_idx.min()) >= 0
    exp = m.select_points(ref_idx.to(DEV), mode='merge')
    got = out['modalities']['image'].mapping
    assert torch.equal(got.pointers, exp.pointers) and torch.equal(got.images, exp.images)
    assert torch.equal(got.pixels, exp.pixels) and torch.allclose(got.features, exp.features)
    assert got.num_groups == out_c.shape[0]
