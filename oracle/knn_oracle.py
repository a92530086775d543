"""CPU restatement of NeighborhoodBasedMappingFeatures (TEST INFRASTRUCTURE ONLY).

Reference: torch_points3d/core/data_transform/multimodal/image.py:431-612.  The module itself cannot be
imported here (its import chain needs torch_geometric.transforms, torch_points_kernels, torch_cluster, ...:
SURVEY.md 8c) and its K-NN is pykeops 1.4.2 `argKmin` (absent) or FAISS (approximate).  Restated:
  * K-NN (:499-508): the k smallest fp32 squared distances ((dx^2 + dy^2) + dz^2) over ALL points, the point
    itself included; ties broken by the lower index (KeOps leaves tie order unspecified -- PARITY UNPINNED for
    tie order only: the k-th distance, hence the density feature, does not depend on it);
  * density (:517-531) and occlusion (:560-599): the reference's expressions, statement by statement.
"""
import numpy as np
import torch


def knn_bruteforce(xyz, k):
    x = np.asarray(xyz, dtype=np.float32)
    n = x.shape[0]
    nbr = np.full((n, k), -1, dtype=np.int32)
    d2o = np.full((n, k), np.inf, dtype=np.float32)
    idx = np.arange(n)
    for i in range(n):
        e = x[i][None, :] - x
        d2 = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]          # fp32, left to right
        order = np.lexsort((idx, d2))[:k]
        nbr[i, :len(order)] = order
        d2o[i, :len(order)] = d2[order]
    return nbr, d2o


def neighborhood_features(xyz, pointers, images, neighbors, k_list, voxel=1, density=True, occlusion=True):
    """Columns appended to the mapping features: densities for every k, then occlusions for every k."""
    xyz = torch.as_tensor(xyz).float()
    pointers = torch.as_tensor(pointers).long()
    images = torch.as_tensor(images).long()
    neighbors = torch.as_tensor(neighbors).long()
    sizes = pointers[1:] - pointers[:-1]
    cols = []
    if density:
        dens = []
        for k in k_list:
            d2_max = ((xyz - xyz[neighbors[:, k - 1]]) ** 2).sum(dim=1)
            v_sphere = 3.1416 * d2_max
            voxel_density = 1 / voxel ** 2
            d = ((k + 1) / v_sphere) / voxel_density
            d[torch.where(d.isnan())] = 1
            dens.append(d.view(-1, 1))
        cols.append(torch.cat(dens, dim=1).repeat_interleave(sizes, 0))
    if occlusion:
        n_points = xyz.shape[0]
        n_images = int(images.max()) + 1
        point_ids = torch.arange(n_points).repeat_interleave(sizes)
        views = torch.zeros((n_points, n_images), dtype=torch.bool)
        views[point_ids, images] = True
        occ = []
        for k in k_list:
            seen = torch.ones_like(images, dtype=torch.float)
            for i in range(k):
                views_neigh = neighbors[:, i].repeat_interleave(sizes)
                seen += views[(views_neigh, images)]
            occ.append((seen / (k + 1)).view(-1, 1))
        cols.append(torch.cat(occ, dim=1))
    return torch.cat(cols, dim=1)
