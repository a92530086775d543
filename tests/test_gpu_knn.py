"""Exact grid K-NN (dva_knn), per-view occlusion (dva_view_occlusion) and the NeighborhoodBasedMappingFeatures
transform (reference core/data_transform/multimodal/image.py:431-612) against oracle/knn_oracle.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden, t
from oracle import knn_oracle as KO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def clouds(name, n, gen):
    if name == "uniform":
        return torch.rand(n, 3, generator=gen) * torch.tensor([4.0, 4.0, 2.5])
    if name == "voxel_grid":      # voxel centres: masses of exactly tied distances
        c = torch.randint(0, 14, (4 * n, 3), generator=gen)
        c = torch.unique(c, dim=0)
        return (c[torch.randperm(c.shape[0], generator=gen)[:n]].float() + 0.5) * 0.05
    if name == "clustered":       # very uneven density + far outliers (many shells)
        a = torch.randn(n - 10, 3, generator=gen) * 0.02
        b = torch.randn(10, 3, generator=gen) * 30
        return torch.cat([a, b])
    raise ValueError(name)


@pytest.mark.parametrize("name", ["uniform", "voxel_grid", "clustered"])
@pytest.mark.parametrize("k", [1, 8, 50])
def test_knn_matches_bruteforce(name, k):
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(k)
    xyz = clouds(name, 1500, gen)
    ref_n, ref_d = KO.knn_bruteforce(xyz.numpy(), k)
    for cell in (None, 0.07, 5.0):
        nbr, d2 = ops.knn(xyz.to(DEV), k, cell=cell)
        assert nbr.dtype == torch.int32
        assert np.array_equal(d2.cpu().numpy(), ref_d), (name, k, cell)          # bit-identical fp32 distances
        assert np.array_equal(nbr.cpu().numpy(), ref_n), (name, k, cell)         # and tie order
    assert torch.equal(nbr[:, 0].cpu().long(), torch.arange(xyz.shape[0]))       # the point itself comes first


def test_knn_fewer_points_than_k_and_empty():
    from deepviewagg_amd import ops
    xyz = torch.rand(5, 3)
    nbr, d2 = ops.knn(xyz.to(DEV), 8)
    ref_n, ref_d = KO.knn_bruteforce(xyz.numpy(), 8)
    assert np.array_equal(nbr.cpu().numpy(), ref_n) and np.array_equal(d2.cpu().numpy(), ref_d)
    assert (nbr[:, 5:] == -1).all() and torch.isinf(d2[:, 5:]).all()
    nbr, d2 = ops.knn(torch.zeros(0, 3, device=DEV), 4)
    assert nbr.shape == (0, 4)


def test_knn_large_against_kdtree():
    """200k surface points: k-th neighbour distances equal those of an exact KD-tree (scipy)."""
    from scipy.spatial import cKDTree
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(0)
    n = 200_000
    face = torch.randint(0, 3, (n,), generator=gen)
    xyz = torch.rand(n, 3, generator=gen) * 4
    xyz[torch.arange(n), face] = (torch.randint(0, 2, (n,), generator=gen).float() * 4)   # box faces
    xyz += torch.randn(n, 3, generator=gen) * 1e-3
    k = 50
    nbr, d2 = ops.knn(xyz.to(DEV), k)
    dist, _ = cKDTree(xyz.numpy().astype(np.float64)).query(xyz.numpy().astype(np.float64), k=k)
    np.testing.assert_allclose(np.sqrt(d2.cpu().numpy().astype(np.float64)), dist, rtol=1e-4, atol=1e-6)
    # the reported distances are those of the reported neighbours
    rec = ((xyz[:, None, :] - xyz[nbr.cpu().long()]) ** 2)
    rec = (rec[..., 0] + rec[..., 1]) + rec[..., 2]
    assert torch.equal(rec, d2.cpu())


def test_neighborhood_features_match_oracle():
    from deepviewagg_amd.core.multimodal.image import ImageMapping
    from deepviewagg_amd.core.data_transform.multimodal.image import NeighborhoodBasedMappingFeatures
    g = load_golden("mapping_build")
    n_pts = len(g["pointers"]) - 1
    m = ImageMapping.from_dense(t(g["dense_point_ids"], DEV), t(g["dense_image_ids"], DEV),
                                t(g["dense_pixels"], DEV), t(g["dense_features"], DEV), num_points=n_pts)
    f0 = m.features.clone()
    gen = torch.Generator().manual_seed(4)
    xyz = torch.rand(n_pts, 3, generator=gen) * 2

    class D:
        pos = xyz

    class I:
        mappings = m
        device = torch.device(DEV)

    k_list = [5, 20]
    tr = NeighborhoodBasedMappingFeatures(k=list(reversed(k_list)), voxel=0.05)
    assert tr.k_list == k_list
    tr(D, I)
    ref_n, _ = KO.knn_bruteforce(xyz.numpy(), k_list[-1])
    exp = KO.neighborhood_features(xyz, m.pointers.cpu(), m.images.cpu(), ref_n, k_list, voxel=0.05)
    got = I.mappings.features.cpu()
    assert got.shape[1] == f0.shape[1] + 4
    assert torch.equal(got[:, :f0.shape[1]], f0.cpu())
    torch.testing.assert_close(got[:, f0.shape[1]:f0.shape[1] + 2], exp[:, :2], rtol=1e-6, atol=0)   # densities
    assert torch.equal(got[:, f0.shape[1] + 2:], exp[:, 2:])                                          # occlusions


@pytest.mark.parametrize("tag", ["surf", "dup"])
def test_neighborhood_features_match_reference_fixture(tag):
    """The HIP transform (dva_knn + dva_view_occlusion + the density expression) against the reference's own
    NeighborhoodBasedMappingFeatures._process output (tests/golden/neighborhood.npz, written by oracle/gen_golden.py
    from core/data_transform/multimodal/image.py:482-612 with a brute-force argKmin in place of KeOps).
    'dup' holds every point twice: zero distances, ties broken towards the lower index on both sides."""
    from deepviewagg_amd.core.multimodal.image import ImageMapping
    from deepviewagg_amd.core.data_transform.multimodal.image import NeighborhoodBasedMappingFeatures
    g = load_golden("neighborhood")
    k_list = [int(k) for k in g["k_list"]]
    xyz = t(g[f"{tag}_xyz"])
    n = xyz.shape[0]
    m = ImageMapping.from_dense(t(g[f"{tag}_point_ids"], DEV), t(g[f"{tag}_image_ids"], DEV),
                                t(g[f"{tag}_pixels"], DEV), t(g[f"{tag}_features_in"], DEV), num_points=n)
    assert torch.equal(m.pointers.cpu(), t(g[f"{tag}_pointers"])) and torch.equal(m.images.cpu(), t(g[f"{tag}_images"]))

    class D:
        pos = xyz
        num_nodes = n

    class I:
        mappings = m
        device = torch.device(DEV)

    NeighborhoodBasedMappingFeatures(k=list(reversed(k_list)), voxel=float(g["voxel"]))(D, I)
    got, ref = I.mappings.features.cpu(), t(g[f"{tag}_features_out"])
    assert got.shape == ref.shape
    assert torch.equal(got[:, :3], ref[:, :3])                                    # the features that were there
    torch.testing.assert_close(got[:, 3:5], ref[:, 3:5], rtol=1e-6, atol=0)       # densities
    assert torch.equal(got[:, 5:], ref[:, 5:])                                    # occlusions
