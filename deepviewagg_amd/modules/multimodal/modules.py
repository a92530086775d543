"""UnimodalBranch: the orchestrator of the multimodal hot path inside a forward pass
(reference: torch_points3d/modules/multimodal/modules.py:249-566).

    IN 3D   ------------------------------------           --  OUT 3D
                                   \\            \\         /
                       Atomic Pool -- View Pool -- Fusion
                     /
    IN Mod  -- Conv -----------------------------------------  OUT Mod

Same constructor, ``forward(mm_data_dict, modality)`` contract and empty-modality behaviour as the
reference; the gather / pooling steps run on the HIP kernels of this package.  In nearest mode the
gather is lazy (``ops.GatheredFeatures``): an exact mapping's atomic pool is the identity, E_mod runs on
the feature-map rows and the gather is fused into the attention kernel.
``MultimodalBlockDown`` wraps the 3D blocks around the branches and keeps the mappings consistent when a
3D block changes the point set (reference modules.py:21-236): sampler-based blocks pick points, strided
sparse convolutions MERGE the mappings of the input voxels onto their parent voxels.  The sparse
convolution itself (torchsparse / MinkowskiEngine) is not part of this package: any block that maps a
voxel tensor exposing ``C`` (int32 [N, 4] coordinates, batch index in the last column), ``s`` (stride) and
``coord_maps`` (stride -> coordinates), i.e. the torchsparse 1.1 SparseTensor interface, can be plugged in;
the parent index is a HIP hash query (``ops.voxel_parent_index``).
"""
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from ... import ops
from ...core.common_modules.base_modules import Identity
from .dropout import ModalityDropout

MODALITY_NAMES = ['image']      # core/multimodal/data.py:10


class SparseVoxels:
    """Minimal voxel tensor with the torchsparse 1.1 ``SparseTensor`` fields the multimodal blocks read
    (``F`` features, ``C`` int32 [N, 4] = x, y, z, batch, ``s`` stride, ``coord_maps`` stride -> coords).
    Lets a ROCm sparse-convolution block (or a test double) sit between the multimodal branches."""

    def __init__(self, F, C, s=1, coord_maps=None):
        self.F, self.C, self.s = F, C, s
        self.coord_maps = coord_maps if coord_maps is not None else {}
        self.coord_maps.setdefault(s, C)


def multimodal_input(data, device, is_multimodal=True):
    """The input dictionary the multimodal blocks consume, from a batch object with ``x``, ``coords``, ``batch``
    and ``modalities`` (reference ``BaseSparseConv3d._set_input``, applications/sparseconv3d.py:145-165):
    ``{'x_3d': voxel tensor, 'x_seen': None, 'modalities': data.to(device).modalities}``; a bare voxel tensor
    for a 3D-only model."""
    from ..SparseConv3d import nn as snn
    x_3d = snn.SparseTensor(data.x, data.coords, data.batch, device)
    if not is_multimodal:
        return x_3d
    return {'x_3d': x_3d, 'x_seen': None, 'modalities': data.to(device).modalities}


def _is_voxel_tensor(x):
    return all(hasattr(x, a) for a in ('C', 's', 'coord_maps'))


class MultimodalBlockDown(nn.Module):
    """ -- 3D Conv ---- Merge i -- 3D Conv --   with one UnimodalBranch per modality in between
    (reference modules.py:21-236).  Built from already-instantiated modules."""

    def __init__(self, block_1, block_2, **kwargs):
        super().__init__()
        self.block_1 = block_1 if block_1 is not None else Identity()
        self.block_2 = block_2 if block_2 is not None else Identity()
        self._modalities = []
        for m, branch in kwargs.items():
            assert m in MODALITY_NAMES, f"Invalid kwarg modality '{m}', expected one of {MODALITY_NAMES}."
            assert isinstance(branch, (UnimodalBranch, IdentityBranch)), \
                f"Expected a UnimodalBranch module for '{m}' modality but got {type(branch)} instead."
            setattr(self, m, branch)
            self._modalities.append(m)
        self.sampler = [getattr(self.block_1, "sampler", None), getattr(self.block_2, "sampler", None)]

    @property
    def modalities(self):
        return self._modalities

    def forward(self, mm_data_dict):
        mm_data_dict = self.forward_3d_block_down(mm_data_dict, self.block_1)
        for m in self.modalities:
            mm_data_dict = getattr(self, m)(mm_data_dict, m)
        return self.forward_3d_block_down(mm_data_dict, self.block_2)

    @staticmethod
    def forward_3d_block_down(mm_data_dict, block):
        """Run a 3D block and re-index ``x_seen`` and every modality's mappings if the block changed the
        point set: i -> idx[i] (reference modules.py:101-236)."""
        if isinstance(block, Identity):
            return mm_data_dict
        x_3d, x_seen = mm_data_dict['x_3d'], mm_data_dict['x_seen']
        idx, mode = None, 'pick'
        if isinstance(x_3d, torch.Tensor):
            # dense features + sampler: the sampler reports which input points were kept
            block.sampler.last_idx = None
            n_in = x_3d.shape[0]
            x_3d = block(x_3d)
            idx_sample = block.sampler.last_idx
            same = idx_sample is not None and idx_sample.shape[0] == n_in and bool(
                (idx_sample == torch.arange(n_in, device=idx_sample.device)).all())
            idx = None if same else idx_sample
        elif _is_voxel_tensor(x_3d):
            mode = 'merge'
            stride_in = x_3d.s
            x_3d = block(x_3d)
            stride_out = x_3d.s
            if stride_in != stride_out:
                assert x_3d.C.shape[1] == 4, \
                    f"Sparse coordinates are expected to have shape (N x 4), with batch indices in the last " \
                    f"column and 3D spatial coordinates in the first ones. Yet, received coordinates tensor " \
                    f"with shape {x_3d.C.shape} instead."
                idx = ops.voxel_parent_index(x_3d.coord_maps[stride_in], x_3d.coord_maps[stride_out],
                                             int(stride_out), batch_col=3)
        else:
            raise NotImplementedError(
                f"Unsupported format for x_3d: {type(x_3d)}. Expected a dense feature tensor or a voxel tensor "
                f"with the torchsparse SparseTensor fields (C, s, coord_maps).")
        if x_seen is not None and idx is not None:
            if mode == 'pick':
                x_seen = x_seen[idx]
            else:
                n_out = x_3d.C.shape[0]
                x_seen = torch.zeros(n_out, dtype=torch.int32, device=x_seen.device).index_add_(
                    0, idx, x_seen.to(torch.int32)).to(x_seen.dtype)
        mm_data_dict['x_3d'], mm_data_dict['x_seen'] = x_3d, x_seen
        for m in mm_data_dict['modalities'].keys():
            mm_data_dict['modalities'][m] = mm_data_dict['modalities'][m].select_points(idx, mode=mode)
        return mm_data_dict


class MultimodalBlockUp(nn.Module):
    """Placeholder of the reference's (empty) up block (modules.py:239-246)."""


class UnimodalBranch(nn.Module):
    def __init__(self, conv, atomic_pool, view_pool, fusion, drop_3d=0, drop_mod=0, hard_drop=False,
                 keep_last_view=False, checkpointing='', out_channels=None, interpolate=False):
        super().__init__()
        self.conv = conv
        self.atomic_pool = atomic_pool
        self.view_pool = view_pool
        self.fusion = fusion
        drop_cls = ModalityDropout if hard_drop else nn.Dropout
        self.drop_3d = drop_cls(p=drop_3d, inplace=False) if drop_3d is not None and drop_3d > 0 else None
        self.drop_mod = drop_cls(p=drop_mod, inplace=True) if drop_mod is not None and drop_mod > 0 else None
        self.keep_last_view = keep_last_view
        self._out_channels = out_channels
        self.interpolate = interpolate
        # 'c' conv, 'a' atomic pooling, 'v' view pooling, 'f' fusion (reference modules.py:283-292)
        assert not checkpointing or isinstance(checkpointing, str), \
            f'Expected checkpointing to be of type str but received {type(checkpointing)} instead.'
        self.checkpointing = ''.join(set('cavf').intersection(set(checkpointing)))

    @property
    def out_channels(self):
        if self._out_channels is None:
            raise ValueError(
                f'{self.__class__.__name__}.out_channels has not been set. Please set it to allow '
                f'inference even when the modality has no data.')
        return self._out_channels

    def forward(self, mm_data_dict, modality):
        x3 = mm_data_dict['x_3d']
        is_sparse_3d = not isinstance(x3, (torch.Tensor, type(None)))
        x_3d = x3.F if is_sparse_3d else x3
        mod_data = mm_data_dict['modalities'][modality]
        is_multi_shape = isinstance(mod_data.x, list)

        def put_x3d(v):
            if is_sparse_3d:
                mm_data_dict['x_3d'].F = v
            else:
                mm_data_dict['x_3d'] = v

        # ---- no modality data at all: emulate the branch output shape (reference :314-365)
        if (is_multi_shape and (len(mod_data) == 0 or all(e.x.shape[0] == 0 for e in mod_data))) \
                or (not is_multi_shape and mod_data.x.shape[0] == 0):
            nc_out, nc_3d = self.out_channels, x_3d.shape[1]
            if nc_out < nc_3d:
                raise ValueError(f'{self.__class__.__name__}.out_channels is smaller than number of '
                                 f'features in x_3d: {nc_out} < {nc_3d}')
            nc_2d = nc_out - nc_3d if nc_out > nc_3d else nc_3d
            if not is_multi_shape:
                mod_data.x = mod_data.x[:, [0]].repeat_interleave(nc_2d, dim=1)
            elif len(mod_data) > 0:
                mod_data.x = [x[:, [0]].repeat_interleave(nc_2d, dim=1) for x in mod_data.x]
            if nc_out > nc_3d:
                zeros = torch.zeros_like(x_3d[:, [0]]).repeat_interleave(nc_2d, dim=1)
                x_3d = torch.cat((x_3d, zeros), dim=1)
            put_x3d(x_3d)
            mm_data_dict['modalities'][modality] = mod_data
            return mm_data_dict

        # ---- some settings are empty: run on the others, then restore (reference :372-393)
        if is_multi_shape and any(e.x.shape[0] == 0 for e in mod_data):
            num = len(mod_data)
            removed = {i: e for i, e in enumerate(mod_data) if e.x.shape[0] == 0}
            indices = [i for i in range(num) if i not in removed]
            mm_data_dict['modalities'][modality] = mod_data[indices]
            mm_data_dict = self.forward(mm_data_dict, modality)
            mod_data = mm_data_dict['modalities'][modality]
            joined = {**{k: e for k, e in zip(indices, mod_data)}, **removed}
            mm_data_dict['modalities'][modality] = mod_data.__class__([joined[i] for i in range(num)])
            return mm_data_dict

        mod_data = self.forward_conv(mod_data)
        x_mod = mod_data.get_mapped_features(interpolate=self.interpolate)
        x_mod = self.forward_atomic_pool(x_3d, x_mod, mod_data.atomic_csr_indexing)
        x_mod, mod_data, csr_idx = self.forward_view_pool(x_3d, x_mod, mod_data)
        x_seen = csr_idx[1:] > csr_idx[:-1]
        x_3d, x_mod, mod_data = self.forward_dropout(x_3d, x_mod, mod_data)
        x_3d = self.forward_fusion(x_3d, x_mod)
        if self._out_channels is None:
            self._out_channels = x_3d.shape[1]
        put_x3d(x_3d)
        mm_data_dict['modalities'][modality] = mod_data
        if mm_data_dict['x_seen'] is None:
            mm_data_dict['x_seen'] = x_seen
        else:
            mm_data_dict['x_seen'] = torch.logical_or(x_seen, mm_data_dict['x_seen'])
        return mm_data_dict

    def forward_conv(self, mod_data, reset=True):
        """2D encoder on each setting's feature maps; the ``x`` setter updates the mapping scale
        (reference modules.py:442-479)."""
        if not self.conv:
            return mod_data
        if isinstance(mod_data.x, list):
            for i in range(len(mod_data)):
                mod_data[i].x = self.forward_conv(mod_data[i], i == 0).x
            return mod_data
        if 'c' in self.checkpointing:
            mod_x = checkpoint(self.conv, mod_data.x.requires_grad_(), torch.BoolTensor([reset]),
                               use_reentrant=True)
        else:
            mod_x = self.conv(mod_data.x, True)
        mod_data.x = mod_x
        return mod_data

    def forward_atomic_pool(self, x_3d, x_mod, csr_idx):
        if isinstance(x_mod, list):
            return [self.forward_atomic_pool(x_3d, x, i) for x, i in zip(x_mod, csr_idx)]
        if 'a' in self.checkpointing and isinstance(x_mod, torch.Tensor):
            return checkpoint(self.atomic_pool, x_3d, x_mod, None, csr_idx, use_reentrant=True)
        return self.atomic_pool(x_3d, x_mod, None, csr_idx)

    def forward_view_pool(self, x_3d, x_mod, mod_data):
        is_multi_shape = isinstance(x_mod, list)
        if is_multi_shape:
            # concatenate the settings' views and bring them into point order (reference :514-525)
            order = mod_data.view_cat_sorting
            if all(isinstance(x, ops.GatheredFeatures) for x in x_mod):
                x_mod = ops.GatheredFeatures.cat(x_mod, order=order)
            elif all(isinstance(x, ops.InterpolatedFeatures) for x in x_mod):
                # bilinear gather of a multi-setting batch (round 6): the settings' taps as ONE lazy gather over the
                # stacked map rows, so the view pooling keeps its fused path (fused_bilinear) instead of [V, C]
                x_mod = ops.InterpolatedFeatures.cat(x_mod, order=order)
            else:
                x_mod = torch.cat([x.materialize() if isinstance(x, ops.LAZY_TYPES) else x
                                   for x in x_mod], dim=0)[order]
            x_map = torch.cat(mod_data.mapping_features, dim=0)[order]
            csr_idx = mod_data.view_cat_csr_indexing
        else:
            x_map = mod_data.mapping_features
            csr_idx = mod_data.view_csr_indexing
        if self.keep_last_view:
            # reference consumers (applications/multimodal/no3d.py:128, view losses) expect the [V, C] tensor
            mod_data.last_view_x_mod = x_mod.materialize() if isinstance(x_mod, ops.LAZY_TYPES) else x_mod
            mod_data.last_view_x_map = x_map
            mod_data.last_view_csr_idx = csr_idx
        if 'v' in self.checkpointing and isinstance(x_mod, torch.Tensor):
            x_mod = checkpoint(self.view_pool, x_3d, x_mod, x_map, csr_idx, use_reentrant=True)
        else:
            x_mod = self.view_pool(x_3d, x_mod, x_map, csr_idx)
        return x_mod, mod_data, csr_idx

    def forward_fusion(self, x_3d, x_mod):
        if 'f' in self.checkpointing:
            return checkpoint(self.fusion, x_3d, x_mod, use_reentrant=True)
        return self.fusion(x_3d, x_mod)

    def forward_dropout(self, x_3d, x_mod, mod_data):
        if self.drop_3d:
            x_3d = self.drop_3d(x_3d)
        if self.drop_mod:
            x_mod = self.drop_mod(x_mod)
            if self.keep_last_view:
                last = mod_data.last_view_x_mod
                last = last.materialize() if isinstance(last, ops.LAZY_TYPES) else last
                mod_data.last_view_x_mod = self.drop_mod(last)
        return x_3d, x_mod, mod_data

    def extra_repr(self) -> str:
        return "\n".join(f'{a}={getattr(self, a)}'
                         for a in ['drop_3d', 'drop_mod', 'keep_last_view', 'checkpointing'])


class IdentityBranch(nn.Module):
    def forward(self, mm_data_dict, modality):
        return mm_data_dict
