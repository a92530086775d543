"""Image-modality data holders and point-image-pixel mappings on the HIP device.

Mirror of the hot-path surface of ``torch_points3d/core/multimodal/image.py`` (reference): the
``ImageMapping`` nested CSR (:1707-2342), ``SameSettingImageData`` (:177-1287), ``ImageData``
(:1409-1595) and their batches, with the names / attributes ``UnimodalBranch`` and the data transforms
poke (SURVEY.md §8b seam 3): ``pointers``, ``values[0]`` images, ``values[1]`` nested CSR of pixels,
``values[2]`` features, ``is_index_value=[True, False, False]``.

Sorting / unique go through the HIP lex kernels (``utils.multimodal``), per-view feature means through
``ops.segment_csr``; what remains is index arithmetic in torch on the device.  File / PIL loading
(``load``, ``read_images``) and the interactive visualisation helpers of the reference are outside the hot
path and not provided.
"""
import copy
from typing import List

import numpy as np
import torch

from ... import ops
from ...utils.multimodal import CompositeTensor, lexargsort, lexargunique, lexunique, tensor_idx
from .csr import CSRBatch, CSRData


def _compute_device(t):
    """Device the HIP kernels run on for tensor ``t`` (CPU inputs are uploaded, like MapImages'
    ``use_cuda`` path in the reference)."""
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        from ..._lib import DvaError
        raise DvaError("ImageMapping construction runs on a HIP device; none is visible "
                       "(deepviewagg_amd has no CPU fallback)")
    return torch.device('cuda', torch.cuda.current_device())


def sparse_interpolation(features, coords, batch, padding_mode='border'):
    """Bilinear interpolation of [B, C, H, W] feature maps at per-row coordinates
    (reference image.py:105-170): ``coords`` [N, 2] = (row, col) in [0, 1] (pixel / (resolution - 1)),
    ``batch`` [N] = which map.  Border-replicate semantics of the reference's default; computed by
    ``dva_gather_bilinear_*``."""
    assert len(features.shape) == 4
    assert coords.shape[0] == batch.shape[0]
    assert len(coords.shape) == 2 and coords.shape[1] == 2
    if padding_mode != 'border':
        raise NotImplementedError(f"padding_mode='{padding_mode}' is not available in the HIP gather "
                                  f"(the reference's default 'border' is)")
    n = batch.shape[0]
    dev = features.device
    packed = ops.pack_gather_index(batch.long(), torch.arange(n + 1, device=dev),
                                   torch.zeros((n, 2), dtype=torch.int16, device=dev))
    return ops.gather_bilinear(features, packed, coords)


# ------------------------------------------------------------------------------------------------
# ImageMapping
# ------------------------------------------------------------------------------------------------

class ImageMapping(CSRData):
    """point -> views -> pixels nested CSR (reference image.py:1707-2342)."""

    @staticmethod
    def from_dense(point_ids, image_ids, pixels, features, num_points=None):
        """Build the mapping from dense (point, image, pixel[, features]) rows
        (reference image.py:1728-1795): stable lexicographic sort by (point, image), views = runs of
        equal (point, image), per-view features = mean over the view's pixels, points without views
        get empty groups."""
        assert point_ids.ndim == 1, 'point_ids and image_ids must be 1D tensors'
        assert point_ids.shape == image_ids.shape, 'point_ids and image_ids must have the same shape'
        assert point_ids.shape[0] == pixels.shape[0], 'pixels and indices must have the same shape'
        assert features is None or point_ids.shape[0] == features.shape[0], \
            'point_ids and features must have the same shape'
        in_device = point_ids.device
        dev = _compute_device(point_ids)
        point_ids, image_ids, pixels = point_ids.to(dev), image_ids.to(dev), pixels.to(dev)
        features = features.to(dev) if features is not None else None

        order = lexargsort(point_ids, image_ids)
        point_ids, image_ids, pixels = point_ids[order], image_ids[order], pixels[order]
        if features is not None:
            features = features[order]
        composite = CompositeTensor(point_ids, image_ids)
        image_pixel_mappings = CSRData(composite.data, pixels, dense=True)
        last = image_pixel_mappings.pointers[1:] - 1
        image_ids, point_ids = image_ids[last], point_ids[last]
        if features is not None:
            feats = features.float() if features.dim() == 2 else features.float().view(-1, 1)
            feats = ops.segment_csr(feats.contiguous(), image_pixel_mappings.pointers, reduce='mean')
            features = feats if features.dim() == 2 else feats.view(-1)
        if features is None:
            mapping = ImageMapping(point_ids, image_ids, image_pixel_mappings, dense=True,
                                   is_index_value=[True, False])
        else:
            mapping = ImageMapping(point_ids, image_ids, image_pixel_mappings, features, dense=True,
                                   is_index_value=[True, False, False])
        top = int(point_ids.max()) + 1
        num_points = top if num_points is None or int(num_points) < top else int(num_points)
        point_ids = point_ids[mapping.pointers[1:] - 1]
        mapping = mapping.insert_empty_groups(point_ids, num_groups=num_points)
        return mapping.to(in_device) if in_device != dev else mapping

    def debug(self):
        super().debug()
        assert len(self.values) == 2 or self.has_features
        assert isinstance(self.values[1], CSRData)
        assert len(self.values[1].values) == 1

    @property
    def points(self):
        return torch.arange(self.num_groups, device=self.device)

    @property
    def images(self):
        return self.values[0]

    @images.setter
    def images(self, images):
        self.values[0] = images.to(self.device)

    @property
    def has_features(self):
        return len(self.values) == 3

    @property
    def features(self):
        return self.values[2] if self.has_features else None

    @features.setter
    def features(self, features):
        if self.has_features:
            if features is None:
                self.values.pop(-1)
            else:
                self.values[2] = features.to(self.device)
        elif features is not None:
            self.values.append(features.to(self.device))

    @property
    def pixels(self):
        return self.values[1].values[0]

    @pixels.setter
    def pixels(self, pixels):
        self.values[1].values[0] = pixels.to(self.device)

    @staticmethod
    def get_batch_type():
        return ImageMappingBatch

    @property
    def num_views(self):
        return self.values[0].shape[0]

    @property
    def num_atoms(self):
        return self.pixels.shape[0]

    @property
    def is_exact(self):
        """Every view owns exactly one pixel (exact=True mapping builds): shapes only, no sync."""
        return self.num_atoms == self.num_views

    def _atom_sizes(self):
        return self.values[1].pointers[1:] - self.values[1].pointers[:-1]

    @property
    def bounding_boxes(self):
        """(w_min, w_max, h_min, h_max) pixel values per image (reference image.py:1859-1869)."""
        image_ids = self.images.repeat_interleave(self._atom_sizes())
        n = int(image_ids.max()) + 1 if image_ids.numel() else 0
        pix = self.pixels.long()
        idx = image_ids.view(-1, 1).expand_as(pix)
        big = torch.iinfo(torch.int64)
        mn = torch.full((n, 2), big.max, dtype=torch.long, device=self.device).scatter_reduce(0, idx, pix, 'amin')
        mx = torch.full((n, 2), big.min, dtype=torch.long, device=self.device).scatter_reduce(0, idx, pix, 'amax')
        return mn[:, 0], mx[:, 0], mn[:, 1], mx[:, 1]

    @property
    def feature_map_indexing(self):
        """Index tuple ``X[idx]`` extracts the mapped features from X [B, C, H, W]
        (reference image.py:1871-1885)."""
        idx_batch = self.images.repeat_interleave(self._atom_sizes())
        return idx_batch.long(), ..., self.pixels[:, 1].long(), self.pixels[:, 0].long()

    @property
    def atomic_csr_indexing(self):
        return self.values[1].pointers

    @property
    def view_csr_indexing(self):
        return self.pointers

    def packed_gather_index(self, ratio=1.0):
        """8-byte (image, x, y) gather index of every atom at feature-map resolution: the HIP
        counterpart of ``rescale_images(1 / ratio).feature_map_indexing``."""
        return ops.pack_gather_index(self.images, self.values[1].pointers, self.pixels, ratio=ratio)

    def rescale_images(self, ratio):
        return self.downscale_images(1 / ratio) if ratio < 1 else self.upscale_images(ratio)

    def downscale_images(self, ratio):
        """Mapping at a lower image resolution: pixels // ratio, duplicate pixels of a view removed
        (reference image.py:1916-1980). Only the atomic (pixel) level changes."""
        assert ratio >= 1, f"Invalid image subsampling ratio: {ratio}. Must be larger than 1."
        out = self.clone()
        if ratio == 1:
            return out
        nested = out.values[1]
        ids = torch.arange(nested.num_items, device=self.device)
        view_ids = torch.arange(nested.num_groups, device=self.device).repeat_interleave(self._atom_sizes())
        pix = nested.values[0]
        pix_x, pix_y = (pix[:, 0] // ratio).long(), (pix[:, 1] // ratio).long()
        # ids are unique, so this keeps every atom, sorted by id (reference :1959)
        keep = lexargunique(ids, pix_x, pix_y) if ids.is_cuda else ids
        view_ids, pix_x, pix_y = view_ids[keep], pix_x[keep], pix_y[keep]
        new_pix = torch.stack((pix_x, pix_y), dim=1).type(pix.dtype)
        if isinstance(nested, CSRBatch):
            sizes = nested.__sizes__
            out.values[1] = CSRBatch(view_ids, new_pix, dense=True)
            out.values[1].__sizes__ = sizes
        else:
            out.values[1] = CSRData(view_ids, new_pix, dense=True)
        return out

    def upscale_images(self, ratio, center=True):
        """Mapping at a higher image resolution (reference image.py:1982-2027)."""
        assert ratio >= 1, f"Invalid image upsampling ratio: {ratio}. Must be larger than 1."
        out = self.clone()
        if ratio == 1:
            return out
        pix = out.pixels
        # one expression, like the reference (image.py:2013-2014): (pix.float() * ratio + ratio / 2).long() -- rounding
        # pix * ratio first differs for non-integer ratios
        new = pix.float() * ratio
        if center:
            new = new + ratio / 2
        out.pixels = new.long().type(pix.dtype)
        return out

    def select_images(self, idx):
        """Keep the mappings to images ``idx`` and renumber them idx[i] -> i
        (reference image.py:2029-2093)."""
        idx = tensor_idx(idx).to(self.device)
        assert idx.unique().numel() == idx.shape[0], "Index must not contain duplicates."
        if self.num_items == 0:
            return self.clone()
        view_idx = torch.where((self.images[..., None] == idx).any(-1))[0]
        values = [val[view_idx] for val in self.values]
        if idx.shape[0] == 0:
            return self.__class__(torch.zeros_like(self.pointers), *values, dense=False,
                                  is_index_value=self.is_index_value)
        idx_gen = torch.full((int(idx.max()) + 1,), -1, dtype=torch.int64, device=self.device)
        idx_gen = idx_gen.scatter_(0, idx, torch.arange(idx.shape[0], device=self.device))
        values[0] = idx_gen[values[0]]
        point_ids = torch.arange(self.num_groups, device=self.device).repeat_interleave(
            self.pointers[1:] - self.pointers[:-1])[view_idx]
        if point_ids.shape[0] == 0:
            return self.__class__(torch.zeros_like(self.pointers), *values, dense=False,
                                  is_index_value=self.is_index_value)
        out = self.__class__(CSRData._sorted_indices_to_pointers(point_ids), *values, dense=False,
                             is_index_value=self.is_index_value)
        point_ids = point_ids[out.pointers[1:] - 1]
        return out.insert_empty_groups(point_ids, num_groups=self.num_groups)

    def crop(self, crop_size, crop_offsets):
        """Copy of the mapping for images cropped to ``crop_size`` (W, H) at per-image ``crop_offsets``
        [B, 2]: pixel coordinates are shifted, pixels outside the box are dropped (reference
        image.py:2279-2342, including its early return when no pixel at all is inside the boxes)."""
        n_img = self.images.unique().numel()
        assert crop_offsets.shape == (n_img, 2), \
            f"Expected crop_offsets to have shape {(n_img, 2)} but got shape {crop_offsets.shape} instead."
        sizes = self._atom_sizes()
        image_ids = self.images.repeat_interleave(sizes)
        pixels = self.pixels - crop_offsets.to(self.device)[image_ids].to(self.pixels.dtype)
        lim = torch.tensor(crop_size, device=self.device)
        inside = torch.where((pixels >= 0).all(dim=1) & (pixels < lim).all(dim=1))[0]
        if inside.shape[0] == 0:
            out = self.clone()
            out.pixels = pixels
            return out
        point_ids = torch.arange(self.num_groups, device=self.device).repeat_interleave(
            self.pointers[1:] - self.pointers[:-1]).repeat_interleave(sizes)
        features = self.features.repeat_interleave(sizes, dim=0)[inside] if self.has_features else None
        return ImageMapping.from_dense(point_ids[inside], image_ids[inside], pixels[inside], features,
                                       num_points=self.num_groups)

    def select_views(self, view_mask):
        """Keep the views selected by a boolean mask; returns (mapping, seen image indices)
        (reference image.py:2095-2165)."""
        assert view_mask.dtype == torch.bool and view_mask.shape[0] == self.num_items
        if self.num_items == 0 or bool(view_mask.all()):
            return self.clone(), None
        view_idx = torch.where(view_mask.to(self.device))[0]
        values = [val[view_idx] for val in self.values]
        point_ids = torch.arange(self.num_groups, device=self.device).repeat_interleave(
            self.pointers[1:] - self.pointers[:-1])[view_idx]
        if view_idx.shape[0] == 0:
            out = self.__class__(torch.zeros_like(self.pointers), *values, dense=False,
                                 is_index_value=self.is_index_value)
            return out, torch.zeros(0, dtype=torch.long, device=self.device)
        seen = torch.unique(values[0])
        values[0] = torch.bucketize(values[0], seen)
        out = self.__class__(CSRData._sorted_indices_to_pointers(point_ids), *values, dense=False,
                             is_index_value=self.is_index_value)
        point_ids = point_ids[out.pointers[1:] - 1]
        return out.insert_empty_groups(point_ids, num_groups=self.num_groups), seen

    def select_points(self, idx, mode='pick'):
        """'pick': keep the points ``idx`` (in that order). 'merge': point i becomes voxel idx[i];
        views of merged points are united, duplicate pixels removed, features of a merged view are
        averaged (reference image.py:2167-2277)."""
        modes = ['pick', 'merge']
        assert mode in modes, f"Unknown mode '{mode}'. Supported modes are {modes}."
        idx = tensor_idx(idx).to(self.device)
        if idx is None or idx.shape[0] == 0 or self.num_groups == 0:
            return self.clone()
        if self.num_items == 0:
            out = self.clone()
            out.pointers = torch.zeros(idx.shape[0] + 1, dtype=torch.long, device=self.device)
            return out
        if mode == 'pick':
            return self[idx]
        if not idx.shape[0] == self.num_groups > 0:
            return self.clone()
        if not torch.arange(int(idx.max()) + 1, device=self.device).equal(idx.unique()):
            return self.clone()
        view_sizes = self.pointers[1:] - self.pointers[:-1]
        atom_sizes = self._atom_sizes()
        point_ids = idx.repeat_interleave(view_sizes)
        image_ids = self.images
        features = None
        if self.has_features:
            features = self.features
            if self.num_items > 1:
                # mean of the features over the views that merge into one (point, image) pair
                view_ids = CompositeTensor(point_ids, image_ids).data
                _, inv = torch.unique(view_ids, return_inverse=True)
                f2 = features.float() if features.dim() > 1 else features.float().view(-1, 1)
                # deterministic: views grouped by a stable sort, summed in their original order by the CSR
                # reduction (an index_add_ on the device is an atomic scatter: the merged features would move by
                # an ulp from call to call)
                order = torch.sort(inv, stable=True).indices
                cnt = torch.bincount(inv, minlength=int(inv.max()) + 1)
                gptr = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)])
                if f2.is_cuda:
                    mean = ops.segment_csr(f2[order].contiguous(), gptr, reduce='mean')[inv]
                else:
                    sums = torch.zeros((cnt.shape[0], f2.shape[1]), dtype=torch.float32).index_add_(0, inv, f2)
                    mean = (sums / cnt.view(-1, 1).float())[inv]
                features = mean if features.dim() > 1 else mean.view(-1)
        point_ids = point_ids.repeat_interleave(atom_sizes)
        image_ids = image_ids.repeat_interleave(atom_sizes)
        if features is not None:
            features = features.repeat_interleave(atom_sizes, dim=0)
        pixels = self.pixels
        keep = lexargunique(point_ids, image_ids, pixels[:, 0], pixels[:, 1])
        return ImageMapping.from_dense(
            point_ids[keep], image_ids[keep], pixels[keep],
            features[keep] if features is not None else None, num_points=int(idx.max()) + 1)


class ImageMappingBatch(ImageMapping, CSRBatch):
    """Batch of ImageMapping (reference image.py:2345-2348)."""
    __csr_type__ = ImageMapping


# ------------------------------------------------------------------------------------------------
# SameSettingImageData
# ------------------------------------------------------------------------------------------------

class SameSettingImageData:
    """Images sharing size / scale / crop settings, with their features ``x`` [B, C, H, W] and the
    ``ImageMapping`` to the 3D points (reference image.py:177-1287)."""

    _numpy_keys = ['path']
    _pinhole_keys = ['fx', 'fy', 'mx', 'my']
    _fisheye_keys = ['xi', 'k1', 'k2', 'gamma1', 'gamma2', 'u0', 'v0']
    _torch_keys = ['pos', 'opk', 'extrinsic', 'crop_offsets', 'rollings'] + _pinhole_keys + _fisheye_keys
    _map_key = 'mappings'
    _x_key = 'x'
    _mask_key = 'mask'
    _visi_key = 'visibility'
    _shared_keys = ['ref_size', 'proj_upscale', 'downscale', 'crop_size', _mask_key, _visi_key]
    _own_keys = _numpy_keys + _torch_keys + [_map_key, _x_key]
    _keys = _shared_keys + _own_keys

    def __init__(self, path=np.empty(0, dtype='O'), pos=torch.empty([0, 3]), opk=None, ref_size=(512, 256),
                 proj_upscale=2, downscale=1, rollings=None, crop_size=None, crop_offsets=None, x=None,
                 mappings=None, mask=None, visibility=None, fx=None, fy=None, mx=None, my=None, xi=None,
                 k1=None, k2=None, gamma1=None, gamma2=None, u0=None, v0=None, extrinsic=None, **kwargs):
        self._x = None
        self._mappings = None
        self.path = np.array(path)
        self.pos = pos.double()
        self.opk = opk.double() if opk is not None else None
        self.fx, self.fy, self.mx, self.my = fx, fy, mx, my
        self.xi, self.k1, self.k2 = xi, k1, k2
        self.gamma1, self.gamma2, self.u0, self.v0 = gamma1, gamma2, u0, v0
        self.extrinsic = extrinsic
        self._ref_size = tuple(ref_size)
        self._proj_upscale = proj_upscale
        self.rollings = rollings if rollings is not None else torch.zeros(self.num_views, dtype=torch.int64)
        self._crop_size = tuple(crop_size) if crop_size is not None else tuple(ref_size)
        self._crop_offsets = (crop_offsets if crop_offsets is not None
                              else torch.zeros((self.num_views, 2), dtype=torch.int64)).to(self.device)
        assert downscale >= 1, f"Expected scalar larger than 1 but got {downscale} instead."
        self._downscale = downscale
        self.x = x
        self.mappings = mappings
        self.mask = mask
        self.visibility = visibility

    def to_dict(self):
        return {key: getattr(self, key) for key in self._keys}

    # -- camera description
    @property
    def num_views(self):
        return self.pos.shape[0]

    @property
    def has_opk(self):
        return getattr(self, 'opk', None) is not None

    @property
    def has_extrinsic(self):
        return getattr(self, 'extrinsic', None) is not None

    @property
    def is_pinhole(self):
        return not any(getattr(self, a, None) is None for a in self._pinhole_keys)

    @property
    def is_fisheye(self):
        return not any(getattr(self, a, None) is None for a in self._fisheye_keys)

    @property
    def is_equirectangular(self):
        return self.has_opk and not self.is_pinhole and not self.is_fisheye

    @property
    def intrinsic_pinhole(self):
        """[B, 4, 4] intrinsic matrices from fx, fy, mx, my (reference image.py:424-439)."""
        if not self.is_pinhole:
            raise ValueError(f"Cannot compute intrinsic matrix, please set {self._pinhole_keys}.")
        k = torch.eye(4).repeat(self.num_views, 1, 1)
        k[:, 0, 0], k[:, 1, 1], k[:, 0, 2], k[:, 1, 2] = self.fx, self.fy, self.mx, self.my
        return k

    @property
    def intrinsic_fisheye(self):
        if not self.is_fisheye:
            raise ValueError(f"Cannot compute intrinsic matrix, please set {self._fisheye_keys}.")
        return torch.stack([self.xi, self.k1, self.k2, self.gamma1, self.gamma2, self.u0, self.v0]).T

    # -- sizes and scales
    @property
    def num_points(self):
        return self.mappings.num_groups if self.mappings is not None else 0

    @property
    def img_size(self):
        """Current (W, H) of ``x`` and of the mappings: crop size / downscale (reference :475-480)."""
        return tuple(int(v / self.downscale) for v in self.crop_size)

    @property
    def ref_size(self):
        return self._ref_size

    @ref_size.setter
    def ref_size(self, ref_size):
        ref_size = tuple(ref_size)
        assert (self.x is None and self.mappings is None) or self._ref_size == ref_size, \
            "Can't edit 'ref_size' if 'x', 'mappings' are not all None."
        assert len(ref_size) == 2
        self._ref_size = ref_size
        self._crop_size = ref_size

    @property
    def pixel_dtype(self):
        """Smallest integer dtype holding the pixel coordinates (reference :507-516)."""
        for dtype in [torch.int16, torch.int32, torch.int64]:
            if torch.iinfo(dtype).max >= max(self.ref_size[0], self.ref_size[1]):
                break
        return dtype

    @property
    def proj_upscale(self):
        return self._proj_upscale

    @proj_upscale.setter
    def proj_upscale(self, scale):
        assert (self.mappings is None and self.mask is None) or self._proj_upscale == scale
        self._proj_upscale = scale

    @property
    def proj_size(self):
        return tuple(int(v * self.proj_upscale) for v in self.ref_size)

    @property
    def crop_size(self):
        return self._crop_size

    @property
    def mapping_size(self):
        return self.crop_size

    @property
    def crop_offsets(self):
        return self._crop_offsets

    @property
    def downscale(self):
        return self._downscale

    @downscale.setter
    def downscale(self, scale):
        assert (self.x is None and self.mappings is None) or self.downscale == scale, \
            "Can't directly edit 'downscale' if 'x' or 'mappings' are not both None. Setting 'x' " \
            "will automatically adjust the scale."
        assert scale >= 1, f"Expected scalar larger than 1 but got {scale} instead."
        self._downscale = scale

    # -- features and mappings
    @property
    def x(self):
        return self._x

    @x.setter
    def x(self, x):
        """Setting the features also updates ``downscale`` from the feature-map size, using the
        largest of the two axis ratios (reference image.py:756-787)."""
        if x is None:
            self._x = None
            return
        assert isinstance(x, torch.Tensor), f"Expected a tensor of image features but got {type(x)} instead."
        assert x.shape[0] == self.num_views, \
            f"Expected a tensor of shape ({self.num_views}, :, {self.img_size[1]}, {self.img_size[0]}) " \
            f"but got {x.shape} instead."
        scale = max(self.img_size[0] / x.shape[3], self.img_size[1] / x.shape[2])
        self._downscale = self.downscale * scale
        self._x = x.to(self.device)

    @property
    def mappings(self):
        return self._mappings

    @mappings.setter
    def mappings(self, mappings):
        if mappings is None:
            self._mappings = None
            return
        assert isinstance(mappings, ImageMapping), f"Expected an ImageMapping but got {type(mappings)} instead."
        self._mappings = mappings.to(self.device)

    @property
    def mask(self):
        return self._mask

    @mask.setter
    def mask(self, mask):
        if mask is not None:
            assert mask.dtype == torch.bool, f"Expected a dtype=torch.bool but got dtype={mask.dtype} instead."
            assert tuple(mask.shape) == self.proj_size, \
                f"Expected mask of size {self.proj_size} but got {tuple(mask.shape)} instead."
            mask = mask.to(self.device)
        self._mask = mask

    def select_points(self, idx, mode='pick'):
        """Reference image.py:826-907: 'pick' also drops the images no selected point sees."""
        idx = tensor_idx(idx).to(self.device)
        if self.mappings is None or idx is None or idx.shape[0] == 0:
            return self.clone()
        if len(self) == 0:
            return self.clone()
        if mode == 'pick':
            mappings = self.mappings.select_points(idx, mode=mode)
            seen = torch.unique(mappings.images) if mappings.num_items > 0 else []
            held, self._mappings = self._mappings, None
            images = self[seen]
            images.mappings = mappings.select_images(seen)
            self._mappings = held
            return images
        if mode == 'merge':
            images = self.clone()
            if not idx.shape[0] == self.num_points > 0:
                return images
            if not torch.arange(int(idx.max()) + 1, device=self.device).equal(idx.unique()):
                return images
            images.mappings = images.mappings.select_points(idx, mode=mode)
            return images
        raise ValueError(f"Unknown point selection mode '{mode}'.")

    def select_views(self, view_mask):
        if self.mappings is None or view_mask is None or bool(torch.all(view_mask)) or len(self) == 0:
            return self.clone()
        mappings, seen = self.mappings.select_views(view_mask)
        held, self._mappings = self._mappings, None
        images = self[seen] if seen is not None else self.clone()
        self._mappings = held
        images.mappings = mappings
        return images

    def update_rollings(self, rollings):
        """Roll spherical images and their mappings along the width, with respect to the reference state
        (reference image.py:578-628).  No prior cropping along the width or resizing."""
        assert self.ref_size[0] == self.img_size[0], \
            "CenterRoll cannot operate if images and mappings underwent prior cropping or resizing."
        assert self.crop_size is None or tuple(self.crop_size) == tuple(self.ref_size), \
            "CenterRoll cannot operate if images and mappings underwent prior cropping or resizing."
        assert self.downscale is None or self.downscale == 1, \
            "CenterRoll cannot operate if images and mappings underwent prior cropping or resizing."
        self.rollings = rollings.to(self.device).long()
        if self.x is not None:
            # per-image roll along W as one gather (the reference loops over images with .item())
            W = self.x.shape[-1]
            src = (torch.arange(W, device=self.device).view(1, -1) - self.rollings.view(-1, 1)) % W
            self.x = torch.gather(self.x, 3, src.view(-1, 1, 1, W).expand_as(self.x))
        if self.mappings is not None:
            pix_roll = self.rollings[self.mappings.images].repeat_interleave(self.mappings._atom_sizes())
            w_pix = (self.mappings.pixels[:, 0].long() + pix_roll) % self.ref_size[0]
            self.mappings.pixels[:, 0] = w_pix.to(self.mappings.pixels.dtype)
        return self

    def update_cropping(self, crop_size, crop_offsets):
        """Crop ``x`` and the mappings with respect to the CURRENT ``img_size``; the stored crop state is
        kept with respect to ``ref_size`` (reference image.py:688-722)."""
        crop_offsets = crop_offsets.long().to(self.device)
        self._crop_size = tuple(int(v * self.downscale) for v in crop_size)
        self._crop_offsets = (self.crop_offsets + crop_offsets * self.downscale).long()
        if self.x is not None:
            Wc, Hc = int(crop_size[0]), int(crop_size[1])
            cols = crop_offsets[:, 0].view(-1, 1) + torch.arange(Wc, device=self.device).view(1, -1)
            rows = crop_offsets[:, 1].view(-1, 1) + torch.arange(Hc, device=self.device).view(1, -1)
            b = torch.arange(self.x.shape[0], device=self.device).view(-1, 1, 1)
            self._x = self.x[b, :, rows.view(-1, Hc, 1), cols.view(-1, 1, Wc)].permute(0, 3, 1, 2).contiguous()
        if self.mappings is not None:
            self.mappings = self.mappings.crop(crop_size, crop_offsets)
        return self

    def __len__(self):
        return self.num_views

    def __getitem__(self, idx):
        """Select images (no duplicates); mappings follow (reference image.py:1109-1148)."""
        idx = tensor_idx(idx).to(self.device)
        assert idx.unique().numel() == idx.shape[0], "Index must not contain duplicates."
        idx_np = np.asarray(idx.cpu())

        def sel(a):
            return a[idx] if a is not None else None
        return self.__class__(
            path=self.path[idx_np], pos=self.pos[idx], opk=sel(self.opk) if self.has_opk else None,
            extrinsic=sel(self.extrinsic) if self.has_extrinsic else None,
            **{k: (sel(getattr(self, k)) if self.is_pinhole else None) for k in self._pinhole_keys},
            **{k: (sel(getattr(self, k)) if self.is_fisheye else None) for k in self._fisheye_keys},
            ref_size=copy.deepcopy(self.ref_size), proj_upscale=copy.deepcopy(self.proj_upscale),
            downscale=copy.deepcopy(self.downscale), crop_size=copy.deepcopy(self.crop_size),
            crop_offsets=self.crop_offsets[idx], x=self.x[idx] if self.x is not None else None,
            mappings=self.mappings.select_images(idx) if self.mappings is not None else None,
            mask=self.mask.clone() if self.mask is not None else None,
            visibility=copy.deepcopy(getattr(self, 'visibility', None)))

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __repr__(self):
        return f"{self.__class__.__name__}(num_views={self.num_views}, num_points={self.num_points}, " \
               f"device={self.device})"

    def clone(self):
        out = copy.copy(self)
        out._x = self.x.clone() if self.x is not None else None
        out._mappings = self.mappings.clone() if self.mappings is not None else None
        return out

    def to(self, device):
        def mv(a):
            return a.to(device) if a is not None else None
        out = self.__class__(
            path=self.path, pos=self.pos.to(device), opk=mv(self.opk) if self.has_opk else None,
            extrinsic=mv(self.extrinsic) if self.has_extrinsic else None,
            **{k: mv(getattr(self, k)) for k in self._pinhole_keys + self._fisheye_keys},
            ref_size=self.ref_size, proj_upscale=self.proj_upscale, downscale=self.downscale,
            rollings=self.rollings.to(device), crop_size=self.crop_size,
            crop_offsets=self.crop_offsets.to(device), visibility=self.visibility)
        out._x = mv(self.x)
        out._mappings = self.mappings.to(device) if self.mappings is not None else None
        out._mask = mv(self.mask)
        return out

    @property
    def device(self):
        return self.pos.device

    @property
    def settings_hash(self):
        keys = tuple(sorted(set(self._shared_keys) - {self._mask_key} - {self._visi_key}))
        return hash(tuple(getattr(self, k) for k in keys))

    @staticmethod
    def get_batch_type():
        return SameSettingImageBatch

    @property
    def feature_map_indexing(self):
        return self.mappings.feature_map_indexing if self.mappings is not None else None

    @property
    def atomic_csr_indexing(self):
        return self.mappings.atomic_csr_indexing if self.mappings is not None else None

    @property
    def view_csr_indexing(self):
        return self.mappings.view_csr_indexing if self.mappings is not None else None

    @property
    def mapping_features(self):
        return self.mappings.features

    def get_mapped_features(self, interpolate=False, lazy=True):
        """Features of the mapped pixels (reference image.py:1262-1287).

        nearest (``interpolate=False`` or feature map at mapping resolution): the mapping is brought to
        the feature-map resolution (``pixels // downscale``) and gathered.  With ``lazy=True`` (default)
        the gather is NOT materialised: an ``ops.GatheredFeatures`` is returned, which the pooling
        modules of this package consume directly (E_mod on the map rows, gather fused into the
        attention kernel); call ``.materialize()`` to obtain the reference's [P, C] tensor.
        bilinear (``interpolate=True`` on a feature map below the mapping resolution): ``sparse_interpolation``
        semantics.  With ``lazy=True`` and an exact mapping (one pixel per view) an ``ops.InterpolatedFeatures`` is
        returned -- the four taps and weights of every view, no [P, C] tensor: ``GroupBimodalCSRPool`` evaluates E_mod
        per view inside its kernels (``fused_bilinear``), everything else calls ``.materialize()``.  ``lazy=False``
        (or a mapping with several pixels per view) returns the reference's [P, C] tensor."""
        scale = 1 / self.downscale
        if interpolate and scale != 1:
            dev = self.device
            resolution = torch.tensor([self.mapping_size], dtype=torch.float32, device=dev)
            coords = (self.mappings.pixels / (resolution - 1))[:, [1, 0]]
            packed = self.mappings.packed_gather_index(ratio=1.0)
            if lazy and self.mappings.is_exact:
                # one pixel per view: the taps are kept, no [P, C] tensor (fused_bilinear through GroupBimodalCSRPool)
                return ops.lazy_gather_bilinear(self.x, packed, coords, exact=True)
            return ops.gather_bilinear(self.x, packed, coords)
        if self.downscale < 1:
            # feature map larger than the mapping resolution: the reference goes through
            # rescale_images -> upscale_images (pix * ratio + ratio / 2, image.py:1982-2027)
            mappings, ratio = self.mappings.upscale_images(1 / self.downscale), 1.0
        else:
            mappings, ratio = self.mappings, float(self.downscale)
        if lazy:
            # (image, pixel) -> map row in one pass: the lazy gather needs no packed index
            return ops.lazy_gather_nearest_mapping(self.x, mappings.images, mappings.values[1].pointers,
                                                   mappings.pixels, ratio, exact=self.mappings.is_exact)
        return ops.gather_nearest(self.x, mappings.packed_gather_index(ratio=ratio))


class SameSettingImageBatch(SameSettingImageData):
    """Batch of SameSettingImageData with the same settings (reference image.py:1290-1406)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.__sizes__ = None

    @property
    def batch_pointers(self):
        return np.cumsum(np.concatenate(([0], self.__sizes__))) if self.__sizes__ is not None else None

    @property
    def batch_items_sizes(self):
        return self.__sizes__

    @property
    def num_batch_items(self):
        return len(self.__sizes__) if self.__sizes__ is not None else None

    @staticmethod
    def from_data_list(image_data_list):
        assert isinstance(image_data_list, list) and len(image_data_list) > 0
        assert all(isinstance(x, SameSettingImageData) for x in image_data_list)
        first = image_data_list[0]
        assert all(im.settings_hash == first.settings_hash for im in image_data_list), \
            f"All SameSettingImageData values for shared keys {SameSettingImageData._shared_keys} must " \
            f"be the same (except for the 'mask')."
        batch_dict = {k: getattr(first, k) for k in SameSettingImageData._shared_keys}
        batch_dict['path'] = np.concatenate([im.path for im in image_data_list])
        for key in SameSettingImageData._torch_keys:
            vals = [getattr(im, key) for im in image_data_list]
            batch_dict[key] = None if any(v is None for v in vals) else torch.cat(vals)
        xs = [im.x for im in image_data_list]
        batch_dict['x'] = None if any(v is None for v in xs) else torch.cat(xs)
        maps = [im.mappings for im in image_data_list]
        batch_dict['mappings'] = None if any(m is None for m in maps) else ImageMappingBatch.from_csr_list(maps)
        batch = SameSettingImageBatch(**batch_dict)
        batch.__sizes__ = np.array([im.num_views for im in image_data_list])
        return batch

    def to_data_list(self):
        if self.__sizes__ is None:
            raise RuntimeError('Cannot reconstruct image data list from batch because the batch object '
                               'was not created using `SameSettingImageBatch.from_data_list()`.')
        bp = self.batch_pointers
        return [self[slice(int(bp[i]), int(bp[i + 1]))] for i in range(self.num_batch_items)]


# ------------------------------------------------------------------------------------------------
# ImageData
# ------------------------------------------------------------------------------------------------

class ImageData:
    """List of SameSettingImageData with different settings; the format multimodal modules work on
    (reference image.py:1409-1595)."""

    def __init__(self, image_list: List[SameSettingImageData]):
        self._list = image_list

    @property
    def num_settings(self):
        return len(self)

    @property
    def num_views(self):
        return sum(im.num_views for im in self)

    @property
    def num_points(self):
        return self[0].num_points if len(self) > 0 else 0

    @property
    def x(self):
        return [im.x for im in self]

    @x.setter
    def x(self, x_list):
        assert x_list is None or isinstance(x_list, list), f"Expected a List but got {type(x_list)} instead."
        if x_list is None or len(x_list) == 0:
            x_list = [None] * self.num_settings
        for im, x in zip(self, x_list):
            im.x = x

    def __len__(self):
        return len(self._list)

    def __getitem__(self, idx):
        if len(self) == 0:
            raise ValueError(f'{self} cannot be indexed because it has length 0.')
        if isinstance(idx, int) and idx < len(self):
            return self._list[idx]
        return self.__class__([self._list[i] for i in tensor_idx(idx).tolist()])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __repr__(self):
        return f"{self.__class__.__name__}(num_settings={self.num_settings}, num_views={self.num_views}, " \
               f"num_points={self.num_points}, device={self.device})"

    def select_points(self, idx, mode='pick'):
        return self.__class__([im.select_points(idx, mode=mode) for im in self])

    def select_views(self, view_mask_list):
        assert isinstance(view_mask_list, list), "Expected a list of view masks."
        return self.__class__([im.select_views(m) for im, m in zip(self, view_mask_list)])

    def clone(self):
        return self.__class__([im.clone() for im in self])

    def to(self, device):
        out = self.clone()
        out._list = [im.to(device) for im in out]
        return out

    @property
    def device(self):
        return self[0].device if len(self) > 0 else 'cpu'

    @staticmethod
    def get_batch_type():
        return ImageBatch

    def get_mapped_features(self, interpolate=False, lazy=True):
        return [im.get_mapped_features(interpolate=interpolate, lazy=lazy) for im in self]

    @property
    def feature_map_indexing(self):
        return [im.feature_map_indexing for im in self]

    @property
    def atomic_csr_indexing(self):
        return [im.atomic_csr_indexing for im in self]

    @property
    def view_cat_sorting(self):
        """Permutation bringing the concatenated per-setting views into point order
        (reference image.py:1549-1574). Stable, so views of one point keep the setting order."""
        dense = torch.cat([
            torch.arange(im.num_points, device=self.device).repeat_interleave(
                im.view_csr_indexing[1:] - im.view_csr_indexing[:-1]) for im in self])
        return torch.sort(dense, stable=True).indices

    @property
    def view_cat_csr_indexing(self):
        """CSR pointers of the concatenated, point-sorted views: sum of the per-setting pointers."""
        return torch.stack([im.view_csr_indexing for im in self], dim=1).sum(dim=1)

    @property
    def mapping_features(self):
        return [im.mapping_features for im in self]


class ImageBatch(ImageData):
    """Batch of ImageData: settings with equal hash are merged into SameSettingImageBatch items and
    the point indices of the mappings are offset (reference image.py:1598-1704)."""

    def __init__(self, image_list):
        super().__init__(image_list)
        self.__il_sizes__ = None
        self.__hashes__ = None
        self.__il_idx_dict__ = None
        self.__im_idx_dict__ = None
        self.__cum_pts__ = None

    @staticmethod
    def from_data_list(image_data_list):
        assert isinstance(image_data_list, list) and len(image_data_list) > 0
        assert all(isinstance(x, ImageData) for x in image_data_list)
        hashes = []
        for il in image_data_list:
            for im in il:
                if im.settings_hash not in hashes:
                    hashes.append(im.settings_hash)
        n_pts = torch.LongTensor([il.num_points for il in image_data_list])
        cum_pts = torch.cumsum(torch.cat((torch.LongTensor([0]), n_pts)), dim=0)
        il_idx_dict = {h: [] for h in hashes}
        im_idx_dict = {h: [] for h in hashes}
        groups = {h: [] for h in hashes}
        for il_idx, il in enumerate(image_data_list):
            for im_idx, im in enumerate(il):
                h = im.settings_hash
                il_idx_dict[h].append(il_idx)
                im_idx_dict[h].append(im_idx)
                groups[h].append(im)
        batches = [SameSettingImageBatch.from_data_list(groups[h]) for h in hashes]
        for h, im in zip(hashes, batches):
            if im.num_points > 0:
                global_idx = torch.cat([torch.arange(int(cum_pts[i]), int(cum_pts[i + 1]))
                                        for i in il_idx_dict[h]])
                im.mappings.insert_empty_groups(global_idx, num_groups=int(cum_pts[-1]))
        out = ImageBatch(batches)
        out.__il_sizes__ = [len(il) for il in image_data_list]
        out.__hashes__ = hashes
        out.__il_idx_dict__ = il_idx_dict
        out.__im_idx_dict__ = im_idx_dict
        out.__cum_pts__ = cum_pts
        return out

    def to_data_list(self):
        assert self.__il_sizes__ is not None, \
            "Cannot reconstruct the list of ImageData because the ImageBatch was not created using " \
            "'ImageBatch.from_data_list'."
        msi_list = [[None] * s for s in self.__il_sizes__]
        for h, ib in zip(self.__hashes__, self):
            for il_idx, im_idx, im in zip(self.__il_idx_dict__[h], self.__im_idx_dict__[h], ib.to_data_list()):
                start, end = int(self.__cum_pts__[il_idx]), int(self.__cum_pts__[il_idx + 1])
                im.mappings = im.mappings[torch.arange(start, end)]
                msi_list[il_idx][im_idx] = im
        return [ImageData(x) for x in msi_list]
