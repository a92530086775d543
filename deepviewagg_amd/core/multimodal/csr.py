"""CSR containers for lists-of-lists stored as tensors (reference: torch_points3d/core/multimodal/csr.py).

``values[k][pointers[i]:pointers[i+1]]`` are the items of group ``i``; ``pointers[0] == 0``.  Values may
themselves be ``CSRData`` (nested CSR: point -> views -> pixels).  Pure index manipulation: works on
whatever device the tensors live on; the reductions / sorts that build the containers are in
``image.py`` / ``utils/multimodal.py`` and run on the HIP device.
"""
import copy

import torch

from ...utils.multimodal import tensor_idx


class CSRData:
    """Pointers + values (reference csr.py:44-302). Subclasses A should define ABatch(A, CSRBatch)
    and return it from ``get_batch_type()``."""

    def __init__(self, pointers, *args, dense=False, is_index_value=None):
        """``dense=True``: ``pointers`` is a sorted tensor of group indices to convert to pointers.
        ``is_index_value[k]``: values[k] holds indices that must be offset when batching."""
        self.pointers = CSRData._sorted_indices_to_pointers(pointers) if dense else pointers
        self.values = [*args] if len(args) > 0 else None
        if is_index_value is None or len(is_index_value) == 0:
            self.is_index_value = torch.zeros(self.num_values, dtype=torch.bool)
        else:
            self.is_index_value = torch.as_tensor(is_index_value, dtype=torch.bool)

    def debug(self):
        """Structural invariants (reference csr.py:81-108)."""
        assert self.pointers[0] == 0, "The first pointer element must always be 0."
        assert torch.all(self.pointers[1:] - self.pointers[:-1] >= 0), "pointer indices must be increasing."
        if self.values is not None:
            assert isinstance(self.values, list), "Values must be held in a list."
            assert all(len(v) == self.num_items for v in self.values), \
                "All value objects must have the same size."
            for v in self.values:
                if isinstance(v, CSRData):
                    v.debug()
            assert self.is_index_value.dtype == torch.bool and self.is_index_value.ndim == 1
            assert self.is_index_value.shape[0] == self.num_values, \
                "is_index_value size must match the number of value tensors."

    def to(self, device):
        out = self.clone()
        out.pointers = out.pointers.to(device)
        for i in range(out.num_values):
            out.values[i] = out.values[i].to(device)
        return out

    def cpu(self):
        return self.to('cpu')

    def cuda(self):
        return self.to('cuda')

    @property
    def device(self):
        return self.pointers.device

    @property
    def num_groups(self):
        return self.pointers.shape[0] - 1

    @property
    def num_values(self):
        return len(self.values) if self.values is not None else 0

    @property
    def num_items(self):
        return int(self.pointers[-1].item())

    @staticmethod
    def get_batch_type():
        return CSRBatch

    def clone(self):
        """Shallow copy: new container, same tensors (reference csr.py:147-156)."""
        out = copy.copy(self)
        out.pointers = copy.copy(self.pointers)
        out.values = copy.copy(self.values)
        return out

    @staticmethod
    def _is_sorted(a):
        return bool(torch.all(a[:-1] <= a[1:]))

    @staticmethod
    def _sorted_indices_to_pointers(indices):
        """Sorted dense group indices -> pointers: a boundary wherever the index changes (csr.py:158-172)."""
        assert indices.dim() == 1, "Only 1D indices are accepted."
        assert indices.shape[0] >= 1, "At least one group index is required."
        assert CSRData._is_sorted(indices), "Indices must be sorted in increasing order."
        dev = indices.device
        return torch.cat([
            torch.zeros(1, dtype=torch.long, device=dev),
            torch.where(indices[1:] > indices[:-1])[0] + 1,
            torch.full((1,), indices.shape[0], dtype=torch.long, device=dev)])

    def reindex_groups(self, group_indices, num_groups=None):
        """Move existing group i to position group_indices[i]; missing positions become empty groups."""
        order = torch.argsort(group_indices)
        return self[order].insert_empty_groups(group_indices[order], num_groups=num_groups)

    def insert_empty_groups(self, group_indices, num_groups=None):
        """In place: existing group i becomes group ``group_indices[i]`` (sorted); indices absent from
        ``group_indices`` become zero-length groups (reference csr.py:197-229)."""
        assert self.num_groups == group_indices.shape[0], \
            "New group indices must correspond to the existing number of groups"
        assert CSRData._is_sorted(group_indices), "New group indices must be sorted."
        g = group_indices.to(self.device)
        top = int(g.max()) + 1
        num_groups = top if num_groups is None else max(top, int(num_groups))
        starts = torch.cat([torch.full((1,), -1, dtype=g.dtype, device=self.device), g])
        ends = torch.cat([g, torch.full((1,), num_groups, dtype=g.dtype, device=self.device)])
        self.pointers = self.pointers.repeat_interleave(ends - starts)
        return self

    @staticmethod
    def _index_select_pointers(pointers, indices):
        """Pointers of the selected groups + the index of their items in the old values
        (reference csr.py:235-264)."""
        assert indices.max() <= pointers.shape[0] - 2
        dev = pointers.device
        sizes = pointers[indices + 1] - pointers[indices]
        pointers_new = torch.cat([torch.zeros(1, dtype=pointers.dtype, device=dev), torch.cumsum(sizes, 0)])
        total = int(pointers_new[-1])
        # item j of new group i comes from old position pointers[indices[i]] + j
        val_idx = torch.arange(total, device=dev) - pointers_new[:-1].repeat_interleave(sizes) \
            + pointers[indices].repeat_interleave(sizes)
        return pointers_new, val_idx

    def __getitem__(self, idx):
        """Select groups with an int / slice / list / array / LongTensor / BoolTensor index."""
        idx = tensor_idx(idx).to(self.device)
        out = self.clone()
        if idx.shape[0] == 0:
            out.pointers = torch.zeros(1, dtype=torch.long, device=self.device)
            out.values = [v[[]] for v in self.values]
        else:
            out.pointers, val_idx = CSRData._index_select_pointers(self.pointers, idx)
            out.values = [v[val_idx] for v in self.values]
        return out

    def __len__(self):
        return self.num_groups

    def __repr__(self):
        info = [f"{k}={getattr(self, k)}" for k in ['num_groups', 'num_items', 'device']]
        return f"{self.__class__.__name__}({', '.join(info)})"


class CSRBatch(CSRData):
    """Several CSRData stacked into one, with the bookkeeping to split them again
    (reference csr.py:305-479)."""
    __csr_type__ = CSRData

    def __init__(self, pointers, *args, dense=False, is_index_value=None):
        super().__init__(pointers, *args, dense=dense, is_index_value=is_index_value)
        self.__sizes__ = None

    @property
    def batch_pointers(self):
        if self.__sizes__ is None:
            return None
        return torch.cumsum(torch.cat((torch.zeros(1, dtype=torch.long), self.__sizes__)), dim=0)

    @property
    def batch_items_sizes(self):
        return self.__sizes__

    @property
    def num_batch_items(self):
        return len(self.__sizes__) if self.__sizes__ is not None else 0

    def to(self, device):
        out = super().to(device)
        out.__sizes__ = self.__sizes__
        return out

    @staticmethod
    def from_csr_list(csr_list):
        assert isinstance(csr_list, list) and len(csr_list) > 0
        assert isinstance(csr_list[0], CSRData), "All provided items must be CSRData objects."
        csr_type = type(csr_list[0])
        assert all(isinstance(c, csr_type) for c in csr_list), "All provided items must have the same class."
        device = csr_list[0].device
        assert all(c.device == device for c in csr_list), "All provided items must be on the same device."
        num_values = csr_list[0].num_values
        assert all(c.num_values == num_values for c in csr_list), \
            "All provided items must have the same number of values."
        is_index_value = csr_list[0].is_index_value
        if is_index_value is not None:
            assert all(bool(torch.equal(c.is_index_value, is_index_value)) for c in csr_list), \
                "All provided items must have the same is_index_value."
        for i in range(num_values):
            assert all(type(c.values[i]) == type(csr_list[0].values[i]) for c in csr_list), \
                "All provided items must have the same value types."
        # pointers: drop each item's leading 0, offset by the items already stacked
        offsets = torch.cumsum(torch.tensor([0] + [c.num_items for c in csr_list[:-1]], dtype=torch.long), 0)
        pointers = torch.cat([torch.zeros(1, dtype=torch.long, device=device)] + [
            c.pointers[1:] + int(off) for c, off in zip(csr_list, offsets)])
        values = []
        for i in range(num_values):
            val_list = [c.values[i] for c in csr_list]
            if isinstance(val_list[0], CSRData):
                val = val_list[0].get_batch_type().from_csr_list(val_list)
            elif is_index_value is not None and bool(is_index_value[i]):
                # index values: offset by (max + 1) of the previous items (reference csr.py:396-402)
                steps = [int(v.max()) + 1 if v.shape[0] > 0 else 0 for v in val_list]
                offs = torch.cumsum(torch.tensor([0] + steps[:-1], dtype=torch.long), 0)
                val = torch.cat([v + int(o) for v, o in zip(val_list, offs)])
            else:
                val = torch.cat(val_list)
            values.append(val)
        batch = csr_type.get_batch_type()(pointers, *values, dense=False, is_index_value=is_index_value)
        batch.__sizes__ = torch.tensor([c.num_groups for c in csr_list], dtype=torch.long)
        batch.__csr_type__ = csr_type
        return batch

    def __getitem__(self, idx):
        """Indexing a batch breaks the from_csr_list / to_csr_list round trip: the result is a plain
        ``__csr_type__`` object (reference csr.py:456-470)."""
        sub = super().__getitem__(idx)
        return self.__csr_type__(sub.pointers, *sub.values, dense=False, is_index_value=sub.is_index_value)

    def __repr__(self):
        info = [f"{k}={getattr(self, k)}" for k in ['num_batch_items', 'num_groups', 'num_items', 'device']]
        return f"{self.__class__.__name__}({', '.join(info)})"

    def to_csr_list(self):
        if self.__sizes__ is None:
            raise RuntimeError(
                'Cannot reconstruct CSRData list from batch because the batch object was not created '
                'using `CSRBatch.from_csr_list()`.')
        group_ptr = self.batch_pointers
        item_ptr = self.pointers[group_ptr.to(self.device)]
        num_batch = self.num_batch_items
        pointers = [self.pointers[int(group_ptr[i]):int(group_ptr[i + 1]) + 1] - item_ptr[i]
                    for i in range(num_batch)]
        values = []
        for i in range(self.num_values):
            v = self.values[i]
            if isinstance(v, CSRBatch):
                values.append(v.to_csr_list())
                continue
            parts = [v[int(item_ptr[j]):int(item_ptr[j + 1])] for j in range(num_batch)]
            if self.is_index_value is not None and bool(self.is_index_value[i]):
                # undo the cumulative (max + 1) offsets
                steps = [int(p.max()) + 1 if p.shape[0] > 0 else 0 for p in parts]
                fixed, off = [], 0
                for p, s in zip(parts, steps):
                    fixed.append(p - off)
                    off = s if p.shape[0] > 0 else off
                parts = fixed
            values.append(parts)
        return [self.__csr_type__(pointers[j], *[values[i][j] for i in range(self.num_values)],
                                  dense=False, is_index_value=self.is_index_value)
                for j in range(num_batch)]
