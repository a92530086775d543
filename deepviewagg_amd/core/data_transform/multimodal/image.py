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

from ...._lib import DvaError
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

    @staticmethod
    def _batch_size(visi_model, n_points, n_images, budget_bytes=None, device=None):
        """Images per visibility batch: what fits ``budget_bytes`` of workspace + outputs (a 2048 x 1024 projection map
        costs 42 MB of z-buffer / pixel maps per image, a candidate 100 bytes).  Default budget: a third of the
        device memory that is free right now, at most 8 GiB (the preprocessing may share the GPU with a training
        process).  ``dva_visibility_batch`` indexes candidates and map pixels of a batch with int32."""
        if budget_bytes is None:
            budget_bytes = 8 << 30
            if torch.cuda.is_available():
                budget_bytes = min(budget_bytes, torch.cuda.mem_get_info(device)[0] // 3)
        w, h = visi_model.img_size
        per_image = w * h * 20 + n_points * 100 + (n_points if getattr(visi_model, 'exact', False)
                                                   else max(n_points, w * h)) * 48
        int32_cap = (2 ** 31 - 1) // max(n_points, w * h, 1)
        return int(max(1, min(n_images, budget_bytes // max(per_image, 1), 64, int32_cap)))

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

        # the reference loops over the images (:238-353); here the images of the setting go through the visibility
        # kernels in batches (VisibilityModel.batch: one set of launches and one host synchronisation per batch) and the
        # per-image post-processing (:294-353) is applied to all rows at once, the image id leading the sort key
        image_ids, point_ids, features, pixels = [], [], [], []
        n_img = images.num_views
        step = self._batch_size(visi_model, xyz.shape[0], n_img, device=xyz.device)

        # DepthBasedVisibility reads one depth map per image: '<...>/depth/<name>_depth.png' next to '<...>/<dir>/<name>_rgb.png'
        # (the S3DIS layout, derived from image.path exactly as the reference does: image.py:262-265)
        extra = {}
        if self.method == 'DepthBasedVisibility' and 'depth_map' not in self.kwargs:
            import os.path as osp

            def depth_paths(sel):
                return [osp.join(osp.dirname(osp.dirname(str(p))), 'depth',
                                 osp.basename(str(p)).replace('_rgb.png', '_depth.png')) for p in images.path[sel]]
            extra['depth_map_path'] = depth_paths

        def run_batch(sel):
            def part(attr):
                return attr[sel].float() if attr is not None else None
            kw = {k: f(sel) for k, f in extra.items()}
            return visi_model.batch(
                xyz, images.pos[sel].float(),
                img_opk=part(images.opk) if images.has_opk else None,
                img_intrinsic_pinhole=images.intrinsic_pinhole[sel].float() if images.is_pinhole else None,
                img_intrinsic_fisheye=images.intrinsic_fisheye[sel].float() if images.is_fisheye else None,
                img_extrinsic=part(images.extrinsic) if images.has_extrinsic else None,
                img_mask=mask, linearity=lin, planarity=pla, scattering=sca, normals=nrm, **kw)
        i0 = 0
        while i0 < n_img:
            sel = slice(i0, min(i0 + step, n_img))
            try:
                out = run_batch(sel)
            except (DvaError, torch.cuda.OutOfMemoryError) as e:
                # the batch does not fit (workspace allocation, or more candidates x images than the kernels index with
                # int32 = DVA_ERR_UNSUPPORTED / DVA_ERR_OVERFLOW): halve it, down to the reference's one image at a time
                # (:238).  Anything else (bad argument, launch error) is a real failure: raise it where it happened
                if step == 1 or (isinstance(e, DvaError) and e.code not in (-2, -4)):
                    raise
                step = max(1, step // 2)
                torch.cuda.empty_cache()
                continue
            i0 = sel.stop
            if out['idx'].shape[0] == 0:
                continue
            i0_batch = sel.start
            img = out['image'] + i0_batch
            pid = point_index[out['idx']]
            off = images.crop_offsets.to(device)[img]
            px = out['x'].long() // images.proj_upscale - off[:, 0]
            py = out['y'].long() // images.proj_upscale - off[:, 1]
            keep = torch.where((px >= 0) & (py >= 0) & (px < images.crop_size[0]) & (py < images.crop_size[1]))
            px, py, pid, img, ft = px[keep], py[keep], pid[keep], img[keep], out['features'].float()[keep]
            px = (px // images.downscale).long()
            py = (py // images.downscale).long()
            if pid.shape[0] == 0:
                continue
            # per image: first occurrence of every (point, pixel) row, sorted by that key (:328)
            u = lexargunique(img, pid, px, py)
            img, pid, ft, pix = img[u], pid[u], ft[u], torch.stack((px[u], py[u]), dim=1).type(images.pixel_dtype)
            present, counts = torch.unique_consecutive(img, return_counts=True)
            for i_image, chunk_p, chunk_f, chunk_x in zip(present.tolist(), pid.split(counts.tolist()),
                                                          ft.split(counts.tolist()), pix.split(counts.tolist())):
                image_ids.append(i_image)
                point_ids.append(chunk_p)
                features.append(chunk_f)
                pixels.append(chunk_x)

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
        assert xyz.shape[0] >= self.k_list[-1], \
            f"NeighborhoodBasedMappingFeatures needs at least k={self.k_list[-1]} points, got {xyz.shape[0]}"
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


# ---------------------------------------------------------------------------------------------------
# Online mapping-selection transforms (reference core/data_transform/multimodal/image.py:615-959).
# Same constructors and `_process(data, images)` contract; the index work stays on the device the mappings
# live on (the reference runs them in CPU DataLoader workers).  `data` is any object with the reference's
# Data fields (`pos`, the MAPPING_KEY attribute, `num_nodes`), accessed as attributes or items.
# ---------------------------------------------------------------------------------------------------

def _get(data, key):
    return data[key] if isinstance(data, dict) else getattr(data, key)


def _set(data, key, value):
    if isinstance(data, dict):
        data[key] = value
    else:
        setattr(data, key, value)


def _num_nodes(data):
    n = data.get('num_nodes') if isinstance(data, dict) else getattr(data, 'num_nodes', None)
    return int(n) if n is not None else int(_get(data, 'pos').shape[0])


class ImageTransform:
    """Transforms on ``(data, images)``; ``images`` may be an ``ImageData`` (list of settings), in which case
    the transform is applied to every ``SameSettingImageData`` (reference :28-63) unless
    ``_PROCESS_IMAGE_DATA``."""
    _PROCESS_IMAGE_DATA = False

    def __call__(self, data, images):
        from ...multimodal.image import ImageData
        if isinstance(images, ImageData) and not self._PROCESS_IMAGE_DATA:
            out = []
            for im in images:
                data, im = self._process(data, im)
                out.append(im)
            return data, ImageData(out)
        return self._process(data, images)

    def __repr__(self):
        attr = ', '.join(f'{k}={v}' for k, v in self.__dict__.items())
        return f'{self.__class__.__name__}({attr})'


class SelectMappingFromPointId(ImageTransform):
    """Keep the mappings of the points listed in ``data[MAPPING_KEY]`` and renumber them 0..n-1 (:615-644)."""

    def __init__(self):
        self.key = MAPPING_KEY

    def _process(self, data, images):
        assert images.mappings is not None
        images = images.select_points(_get(data, self.key).to(images.device), mode='pick')
        _set(data, self.key, torch.arange(_num_nodes(data), device=images.device))
        return data, images


class DropImagesOutsideDataBoundingBox(ImageTransform):
    """Drop the images whose position is outside the bounding box of the points (:647-667)."""

    def __init__(self, margin=0, ignore_z=False):
        self.margin = margin
        self.ignore_z = ignore_z

    def _process(self, data, images):
        pos = _get(data, 'pos').to(images.device)
        b_min = pos.min(dim=0).values - self.margin / 2
        b_max = pos.max(dim=0).values + self.margin / 2
        mask = torch.logical_and(b_min < images.pos, images.pos < b_max)
        mask = mask[:, 0] * mask[:, 1] if self.ignore_z else mask[:, 0] * mask[:, 1] * mask[:, 2]
        return data, images[mask]


class PickKImages(ImageTransform):
    """K random images, or one image out of K in their order (:689-710)."""

    def __init__(self, k, random=False, replace=False):
        self.k = k
        self.random = random
        self.replace = replace

    def _process(self, data, images):
        if self.random:
            import numpy as np
            idx = torch.from_numpy(np.random.choice(range(images.num_views), size=self.k, replace=self.replace))
        else:
            idx = slice(0, images.num_views, self.k)
        return data, images[idx]


class PickImagesFromMappingArea(ImageTransform):
    """Keep the (at most n_max) images whose mappings cover more than ``area_ratio`` of the image, largest
    first (:713-762)."""

    def __init__(self, area_ratio=0.02, n_max=None, n_min=0, use_bbox=False):
        self.area_ratio = area_ratio
        self.n_max = n_max if n_max is not None and n_max >= 1 else None
        self.n_min = n_min if n_max is not None and n_min >= 0 else 0
        self.use_bbox = use_bbox

    def _process(self, data, images):
        assert images.mappings is not None, "No mappings found in images."
        m = images.mappings
        threshold = images.img_size[0] * images.img_size[1] * self.area_ratio
        atom_ptr = m.values[1].pointers
        pixel_idx = m.images.repeat_interleave(atom_ptr[1:] - atom_ptr[:-1])
        B = images.num_views
        if not self.use_bbox:
            areas = torch.zeros(B, device=pixel_idx.device).index_add_(
                0, pixel_idx, torch.ones(pixel_idx.shape[0], device=pixel_idx.device))
        else:
            pix = m.pixels.int()
            big = torch.iinfo(torch.int32).max
            lo = torch.full((B, 2), big, dtype=torch.int32, device=pix.device)
            hi = torch.full((B, 2), -big, dtype=torch.int32, device=pix.device)
            lo.scatter_reduce_(0, pixel_idx.view(-1, 1).expand(-1, 2), pix, 'amin')
            hi.scatter_reduce_(0, pixel_idx.view(-1, 1).expand(-1, 2), pix, 'amax')
            empty = lo[:, 0] == big                       # torch_scatter leaves 0 for images without pixels
            areas = ((hi[:, 0] - lo[:, 0]) * (hi[:, 1] - lo[:, 1])).masked_fill(empty, 0)
        n_max = images.num_views if self.n_max is None else self.n_max
        idx = areas.argsort().flip(0)
        idx = idx[areas[idx] > threshold][:n_max]
        if idx.shape[0] == 0 and images.num_views > 0 and self.n_min > 0:
            idx = idx[:self.n_min]
        return data, images[idx]


class PickMappingsFromMappingFeatures(ImageTransform):
    """Drop the views whose mapping features are outside (lower, upper) bounds (:877-931)."""

    def __init__(self, feat=None, lower=None, upper=None):
        self.feat = self.sanitize(feat)
        self.lower = self.sanitize(lower)
        self.upper = self.sanitize(upper)
        if len(self.lower) == 0:
            self.lower = [None] * len(self.feat)
        if len(self.upper) == 0:
            self.upper = [None] * len(self.feat)
        for x in [self.lower, self.upper]:
            assert len(x) == len(self.feat), f"{x} has {len(x)} elements but {len(self.feat)} were expected."

    @staticmethod
    def sanitize(x):
        if x is None:
            return []
        return list(x) if isinstance(x, (list, tuple)) else [x]

    def _process(self, data, images):
        if images.mappings is None or not images.mappings.has_features or len(self.feat) == 0:
            return data, images
        m = images.mappings
        assert max(self.feat) == 0 or max(self.feat) < m.features.shape[1], \
            f"Out of bounds feature id {max(self.feat)}."
        features = m.features.view(m.num_items, -1)
        view_mask = torch.ones(m.num_items, dtype=torch.bool, device=features.device)
        for i_feat, lower, upper in zip(self.feat, self.lower, self.upper):
            if lower is not None:
                view_mask = view_mask & (features[:, i_feat] > lower)
            if upper is not None:
                view_mask = view_mask & (features[:, i_feat] < upper)
        return data, images.select_views(view_mask)


class JitterMappingFeatures(ImageTransform):
    """Clamped gaussian noise on the mapping features (:934-959)."""

    def __init__(self, sigma=0.02, clip=0.03):
        self.sigma = sigma
        self.clip = clip

    def _process(self, data, images):
        if images.mappings is None or not images.mappings.has_features:
            return data, images
        f = images.mappings.features
        noise = (self.sigma * torch.randn(f.shape, device=f.device)).clamp(-self.clip, self.clip)
        images.mappings.features = f + noise
        return data, images


class PickImagesFromMemoryCredit(ImageTransform):
    """Cherry-pick images of an ``ImageData`` under a pixel-memory credit, optionally favouring images that
    see yet-unseen points (:765-874).  Same sequential sampling (``np.random.choice``) as the reference."""
    _PROCESS_IMAGE_DATA = True

    def __init__(self, credit=None, img_size=[], k_coverage=0, n_img=0):
        if credit is not None:
            self.credit = credit
        elif len(img_size) == 2 and n_img > 0:
            self.credit = img_size[0] * img_size[1] * n_img
        else:
            raise ValueError("Either credit or img_size and n_img must be provided.")
        self.use_coverage = k_coverage > 0
        self.k_coverage = k_coverage

    def _process(self, data, images):
        import numpy as np
        from ...multimodal.image import ImageData
        if images.num_views == 0:
            return data, images
        # flat table of candidate images: (setting, index in setting, pixel area)
        setting = np.concatenate([np.full(im.num_views, i) for i, im in enumerate(images)])
        local = np.concatenate([np.arange(im.num_views) for im in images])
        area = np.array([images[int(i)].img_size[0] * images[int(i)].img_size[1] for i in setting], dtype=np.float64)
        unseen = None
        if self.use_coverage:
            # unseen[i, p]: image i sees point p and no picked image has seen p yet
            seen = torch.zeros(len(setting), _num_nodes(data), dtype=torch.bool, device=images.device)
            first = 0
            for im in images:
                mp = im.mappings
                pts = torch.arange(mp.num_groups, device=seen.device).repeat_interleave(mp.pointers[1:] - mp.pointers[:-1])
                seen[mp.images + first, pts] = True
                first += im.num_views
            unseen = seen.cpu().numpy()
        credit = self.credit
        assert credit > 0 and credit >= area.min(), \
            f"Insufficient credit={credit} to pick any of the provided images with min_size={int(area.min())}."
        alive = np.ones(len(setting), dtype=bool)
        chosen = []
        while credit > 0 and alive.any():
            alive &= area <= credit                         # images that no longer fit the remaining credit
            cand = np.flatnonzero(alive)
            if cand.size == 0:
                break
            if self.use_coverage:
                cover = unseen[cand].sum(axis=1)
                w_cov = self.k_coverage * cover / (cover.max() + 1)
            else:
                w_cov = 0.0
            weights = area[cand] / area[cand].max() + w_cov      # size weight + coverage weight (:836-850)
            pick = cand[np.random.choice(cand.size, p=weights / weights.sum())]
            chosen.append(pick)
            alive[pick] = False
            credit -= area[pick]
            if self.use_coverage:
                unseen &= ~unseen[pick]
        groups = []
        for i, im in enumerate(images):
            mine = [int(local[c]) for c in chosen if setting[c] == i]
            if mine:
                groups.append(im[torch.LongTensor(mine)])
        return data, ImageData(groups)


class CenterRoll(ImageTransform):
    """Roll spherical images and mappings along the width so that the mappings sit as close to the image
    centre as possible (reference :962-1037): per image, the roll offset (among ``angular_res`` candidates)
    minimising span + centring distance of the mapped columns, evaluated in 8-bit angular coordinates."""

    def __init__(self, angular_res=16):
        assert isinstance(angular_res, int)
        assert angular_res <= 256
        self.angular_res = angular_res

    def _process(self, data, images):
        from ....utils.multimodal import lexunique
        assert images.mappings is not None, "No mappings found in images."
        name = self.__class__.__name__
        assert images.ref_size[0] == images.img_size[0], \
            f"{name} cannot operate if images and mappings underwent prior cropping or resizing."
        assert images.crop_size is None or images.crop_size[0] == images.ref_size[0], \
            f"{name} cannot operate if images and mappings underwent prior cropping or resizing."
        assert images.downscale is None or images.downscale == 1, \
            f"{name} cannot operate if images and mappings underwent prior cropping or resizing."
        m = images.mappings
        if m.images.shape[0] == 0:
            return data, images
        dev = m.device
        B, W = images.num_views, images.ref_size[0]
        # mapped columns in 8-bit angular coordinates, one entry per distinct (image, column)
        img, col8 = lexunique(m.images.repeat_interleave(m._atom_sizes()), (m.pixels[:, 0].float() * 256 / W).long())
        assert torch.equal(torch.unique(img), torch.arange(B, device=dev)), \
            "Image indices discrepancy in the rollings."
        step = int(256 / self.angular_res)
        cand = torch.arange(0, 256, step, device=dev)                       # candidate roll offsets
        rolled = (col8.view(-1, 1) + cand.view(1, -1)) % 256                  # uint8 wrap-around, [pixels, cand]
        tgt = img.view(-1, 1).expand_as(rolled)
        lo = torch.full((B, cand.numel()), 255, dtype=torch.long, device=dev).scatter_reduce_(0, tgt, rolled, 'amin')
        hi = torch.zeros((B, cand.numel()), dtype=torch.long, device=dev).scatter_reduce_(0, tgt, rolled, 'amax')
        # cost = angular span + distance of the span's centre to the image centre (:1011-1020)
        cost = (hi - lo).int() + ((hi.float() + lo) / 2. - 128).abs().int()
        best = cand[cost.min(dim=1).indices]
        rollings = (best.float() / 256. * W).long()
        images.update_rollings(rollings)
        return data, images


class CropImageGroups(ImageTransform):
    """Greedy cropping of the images around their mappings (+ padding) into a family of power-of-two crop
    sizes; images of the same crop size are grouped (reference :1040-1141).  Returns an ``ImageData`` with
    one ``SameSettingImageData`` per crop size."""

    def __init__(self, padding=0, min_size=64):
        assert padding >= 0, f"Expected a positive scalar but got {padding} instead."
        assert ((min_size & (min_size - 1)) == 0) & (min_size != 0), \
            f"Expected a power of two but got {min_size} instead."
        self.padding = padding
        self.min_size = min_size

    def _process(self, data, images):
        from ...multimodal.image import ImageData
        assert images.mappings is not None, "No mappings found in images."
        if images.num_views == 0:
            return data, ImageData([images])
        dev = images.device
        W, H = images.img_size
        x0, x1, y0, y1 = images.mappings.bounding_boxes
        x0, y0 = (x0 - self.padding).clamp(min=0), (y0 - self.padding).clamp(min=0)
        x1, y1 = (x1 + self.padding).clamp(0, W), (y1 + self.padding).clamp(0, H)
        bw, bh = x1 - x0, y1 - y0
        # the ladder of crop sizes: (s, s), (2s, s), (2s, 2s), (4s, 2s), ... capped by the image, ending with
        # the full image (:1085-1123)
        ladder, size, k = [], (self.min_size, self.min_size), 0
        while size[0] <= W and size[1] <= H and size != (W, H):
            ladder.append(size)
            size = (min(size[0] * 2 ** ((k + 1) % 2), W), min(size[1] * 2 ** (k % 2), H))
            k += 1
        ladder.append((W, H))
        # every image goes to the first size of the ladder that holds its padded bounding box
        fits = torch.stack([(bw <= sw) & (bh <= sh) for sw, sh in ladder[:-1]] +
                           [torch.ones_like(bw, dtype=torch.bool)], dim=1)
        level = fits.int().argmax(dim=1)
        crop_families = {}
        for lv in torch.unique(level).tolist():
            sw, sh = ladder[lv]
            idx = torch.where(level == lv)[0]
            # centre the box in the crop, inside the image borders
            off_x = (x0[idx] - (sw - bw[idx]) / 2.).long().clamp(0, W - sw)
            off_y = (y0[idx] - (sh - bh[idx]) / 2.).long().clamp(0, H - sh)
            crop_families[(sw, sh)] = images[idx].update_cropping((sw, sh), torch.stack((off_x, off_y), dim=1))
        return data, ImageData(list(crop_families.values()))


# ---- transforms on the raw images that touch the mappings or the feature layout (reference :1163-1232) -----------

class AddPixelHeightFeature(ImageTransform):
    """Append the normalised pixel height as an image channel (:1163-1176)."""

    def _process(self, data, images):
        b, _, h, w = images.x.shape
        feat = torch.linspace(0, 1, h, device=images.x.device).float().view(1, 1, h, 1).repeat(b, 1, 1, w)
        images.x = torch.cat((images.x, feat), 1)
        return data, images


class AddPixelWidthFeature(ImageTransform):
    """Append the normalised pixel width as an image channel (:1179-1192)."""

    def _process(self, data, images):
        b, _, h, w = images.x.shape
        feat = torch.linspace(0, 1, w, device=images.x.device).float().view(1, 1, 1, w).repeat(b, 1, h, 1)
        images.x = torch.cat((images.x, feat), 1)
        return data, images


class RandomHorizontalFlip(ImageTransform):
    """Flip images AND the mapped pixel columns with probability ``p`` (:1195-1218); the draw is the reference's
    ``torch.rand(1) <= p`` on the host generator."""

    def __init__(self, p=0.50):
        self.p = p

    def _process(self, data, images):
        if torch.rand(1) <= self.p:
            images.x = torch.flip(images.x, [3])
            width = images.x.shape[-1]
            pix = images.mappings.pixels
            pix[:, 0] = (width - 1 - pix[:, 0].long()).to(pix.dtype)
        return data, images


class ToFloatImage(ImageTransform):
    """[0, 255] uint8 images -> [0, 1] float tensors (:1221-1232)."""

    def _process(self, data, images):
        images.x = images.x.float() / 255
        return data, images
