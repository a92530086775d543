"""MapImages: build the point-image-pixel mappings of a point cloud on the HIP device.

Mirror of ``MapImages._process`` (reference: torch_points3d/core/data_transform/multimodal/image.py:
162-428).  Per image: visibility model -> pixel coordinates brought to the reference image frame
(``// proj_upscale``, crop offsets, in-crop filter, ``// downscale``, :307-319) -> duplicate
(point, pixel) rows removed (:328) ; then unseen images are dropped and renumbered (:392-394) and
the global ``ImageMapping`` is assembled with ``from_dense`` (:415-417).

Candidate points: the reference first restricts the cloud to a sphere / cylinder of radius r_max
around the camera with a KD-tree (:242-245), in the tree's (unspecified) order.  The visibility
kernel applies the same ``dist < r_max`` cull itself, so here every point is a candidate, in the
cloud's own order (one admissible KD-tree order; tie-breaks depend on it, see visibility.py).
"""
import torch

from ....utils.multimodal import MAPPING_KEY, lexargunique
from ...multimodal import visibility as visibility_module
from ...multimodal.image import ImageMapping, SameSettingImageData


class MapImages:
    def __init__(self, method='SplattingVisibility', proj_upscale=None, ref_size=None, use_cuda=True,
                 verbose=False, cylinder=False, **kwargs):
        self.key = MAPPING_KEY
        self.verbose = verbose
        self.cylinder = cylinder
        self.ref_size = ref_size
        self.proj_upscale = proj_upscale
        self.method = method
        self.use_cuda = True   # the mapping build only exists on the HIP device
        self.kwargs = kwargs

    def __call__(self, data, images):
        return self._process(data, images)

    def _process(self, data, images: SameSettingImageData):
        assert hasattr(data, self.key)
        assert isinstance(images, SameSettingImageData)
        assert images.num_views >= 1, "At least one image must be provided."
        in_device = images.device
        device = torch.device('cuda', torch.cuda.current_device())
        if self.ref_size is not None:
            images.ref_size = self.ref_size
        if self.proj_upscale is not None:
            images.proj_upscale = self.proj_upscale
        if images.mask is not None:
            assert tuple(images.mask.shape) == images.proj_size
        visi_cls = getattr(visibility_module, self.method)
        visi_model = visi_cls(img_size=images.proj_size, **self.kwargs)

        def dev32(name):
            a = getattr(data, name, None)
            return a.float().to(device) if a is not None else None
        xyz = data.pos.float().to(device)
        point_index = getattr(data, self.key).to(device)
        lin, pla, sca, nrm = dev32('linearity'), dev32('planarity'), dev32('scattering'), dev32('norm')
        mask = images.mask.to(device) if images.mask is not None else None

        image_ids, point_ids, features, pixels = [], [], [], []
        for i_image in range(images.num_views):
            def one(attr):
                return attr[i_image].squeeze().float() if attr is not None else None
            out = visi_model(
                xyz, images.pos[i_image].squeeze().float(),
                img_opk=one(images.opk) if images.has_opk else None,
                img_intrinsic_pinhole=images.intrinsic_pinhole[i_image].float() if images.is_pinhole else None,
                img_intrinsic_fisheye=images.intrinsic_fisheye[i_image].float() if images.is_fisheye else None,
                img_extrinsic=one(images.extrinsic) if images.has_extrinsic else None,
                img_mask=mask, linearity=lin, planarity=pla, scattering=sca, normals=nrm)
            if out['idx'].shape[0] == 0:
                continue
            pid = point_index[out['idx']]
            off = images.crop_offsets[i_image].to(device)
            px = out['x'].long() // images.proj_upscale - off[0]
            py = out['y'].long() // images.proj_upscale - off[1]
            keep = torch.where((px >= 0) & (py >= 0) & (px < images.crop_size[0]) & (py < images.crop_size[1]))
            px, py, pid, ft = px[keep], py[keep], pid[keep], out['features'].float()[keep]
            px = (px // images.downscale).long()
            py = (py // images.downscale).long()
            if pid.shape[0] == 0:
                continue
            u = lexargunique(pid, px, py)
            image_ids.append(i_image)
            point_ids.append(pid[u])
            features.append(ft[u])
            pixels.append(torch.stack((px[u], py[u]), dim=1).type(images.pixel_dtype))

        if len(image_ids) == 0:
            raise ValueError(
                "No mappings were found between the 3D points and any of the provided images. This will "
                "cause errors in the subsequent operations. Make sure your images are located in the "
                "vicinity of your point cloud and that the projection parameters allow for at least one "
                "point-image-pixel mapping before re-running this transformation.")
        seen = torch.tensor(image_ids, dtype=torch.long)
        images = images[seen.to(in_device)]
        dense_ids = torch.arange(len(image_ids), device=device).repeat_interleave(
            torch.tensor([p.shape[0] for p in point_ids], device=device))
        mappings = ImageMapping.from_dense(
            torch.cat(point_ids), dense_ids, torch.cat(pixels), torch.cat(features),
            num_points=int(getattr(data, self.key).max()) + 1)
        images.mappings = mappings.to(in_device)
        images.visibility = visi_model
        return data, images

    def __repr__(self):
        return f"{self.__class__.__name__}(method={self.method}, ref_size={self.ref_size}, " \
               f"proj_upscale={self.proj_upscale}, kwargs={self.kwargs})"


class NeighborhoodBasedMappingFeatures:
    """Append neighbourhood-based mapping features to ``images.mappings.features`` (reference
    core/data_transform/multimodal/image.py:431-612): per k in ``k`` a *density* column (surface density
    from the distance to the k-th neighbour, normalised by ``voxel``) and an *occlusion* column (ratio of
    the k nearest neighbours seen by the same image).  Same constructor as the reference; the K-NN search
    is the exact HIP grid search of ``ops.knn`` (the reference's KeOps path is exact too; its FAISS path is
    approximate), so ``use_cuda / use_faiss / ncells / nprobes`` are accepted and ignored."""

    def __init__(self, k=20, voxel=None, density=True, occlusion=True, use_cuda=False, use_faiss=True, ncells=None,
                 nprobes=10, verbose=False):
        self.k_list = sorted(k) if isinstance(k, list) else [k]
        self.voxel = voxel if voxel is not None else 1
        self.compute_density = density
        self.compute_occlusion = occlusion
        self.verbose = verbose
        assert density or occlusion, "At least one of `density` or `occlusion` must be True."

    def __call__(self, data, images):
        return self._process(data, images)

    def _process(self, data, images):
        from .... import ops
        assert images.mappings is not None
        in_device = images.device
        device = torch.device('cuda', torch.cuda.current_device())
        xyz = data.pos.float().to(device)
        mappings = images.mappings
        pointers = mappings.pointers.to(device)
        neighbors, _ = ops.knn(xyz, self.k_list[-1])
        new = []
        if self.compute_density:
            cols = []
            for k in self.k_list:
                # farthest neighbour of the k-neighbourhood, surface density of the disk of that radius
                # (:522-531: same expressions, same constants)
                d2_max = ((xyz - xyz[neighbors[:, k - 1].long()]) ** 2).sum(dim=1)
                v_sphere = 3.1416 * d2_max
                density = ((k + 1) / v_sphere) / (1 / self.voxel ** 2)
                density[torch.where(density.isnan())] = 1
                cols.append(density.view(-1, 1))
            dens = torch.cat(cols, dim=1)
            new.append(dens.repeat_interleave(pointers[1:] - pointers[:-1], 0))
        if self.compute_occlusion:
            image_ids = mappings.images.to(device)
            n_images = int(image_ids.max()) + 1 if image_ids.numel() > 0 else 1
            new.append(ops.view_occlusion(pointers, image_ids, neighbors, self.k_list, n_images))
        new = torch.cat(new, dim=1).to(in_device)
        if not mappings.has_features:
            mappings.features = new
        else:
            mappings.features = torch.cat([mappings.features, new], dim=1)
        return data, images
