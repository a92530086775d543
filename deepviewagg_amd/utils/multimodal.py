"""Lexicographic sort / unique primitives on composite integer keys, on the HIP device.

Mirror of ``torch_points3d/utils/multimodal.py`` (reference :10-94, :97-179): same names and
argument meaning.  The composite key is built with torch ops on the device; sorting and
first-occurrence selection run in ``dva_argsort_i64`` / ``dva_argunique_i64`` (stable radix sort,
so results are deterministic and ``lexargunique`` equals numpy's ``unique(return_index=True)``).
Inputs must live on a HIP device (no CPU fallback in this package).
"""
import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, require_device, stream_of

# Key under which per-point mapping indices are stored; must contain 'index' so that
# torch_geometric's Batch.from_data_list offsets it when stacking (reference :6-10).
MAPPING_KEY = 'mapping_index'


def tensor_idx(idx):
    """Convert an int, slice, list, numpy index or bool mask to a LongTensor (reference :13-33)."""
    if idx is None:
        idx = torch.LongTensor([])
    elif isinstance(idx, int):
        idx = torch.LongTensor([idx])
    elif isinstance(idx, list):
        idx = torch.LongTensor(idx)
    elif isinstance(idx, slice):
        idx = torch.arange(idx.stop)[idx]
    elif isinstance(idx, np.ndarray):
        idx = torch.from_numpy(idx)
    if idx.dtype == torch.bool:
        idx = torch.where(idx)[0]
    assert idx.dtype is torch.int64, f"Expected LongTensor but got {idx.dtype} instead."
    return idx


class CompositeTensor:
    """Combine 1D int/bool tensors of equal shape into one int64 key that preserves their
    lexicographic order: key = sum_i a_i * prod_{j>i}(max_j + 1) (reference :97-179)."""

    _SUPPORTED = (torch.int8, torch.int16, torch.int32, torch.int64, torch.bool)

    def __init__(self, *args, device=None):
        assert len(args) > 0, "At least one tensor must be provided."
        tensors = [torch.from_numpy(a) if isinstance(a, np.ndarray) else a for a in args]
        if device is not None:
            tensors = [a.to(device) for a in tensors]
        assert tensors[0].ndim == 1, 'Only 1D tensors are accepted as input.'
        assert all(a.shape == tensors[0].shape for a in tensors), \
            'All input tensors must have the same shape.'
        assert all(a.dtype in self._SUPPORTED for a in tensors), \
            f'All input tensors must be in {self._SUPPORTED}. Received types: {[a.dtype for a in tensors]}'
        self.dtype_list = [a.dtype for a in tensors]
        self.dtype = torch.int64
        if tensors[0].shape[0] == 0:
            max_list = [0] * len(tensors)
        else:
            # one host sync for all maxima (the reference syncs once per tensor, :139)
            max_list = (torch.stack([a.long().abs().max() for a in tensors]) + 1).tolist()
        prod = 1
        for m in max_list:
            prod *= int(m)
        assert prod < torch.iinfo(torch.int64).max, \
            'The dtype of at least one of the input tensors must allow the composite computation.'
        self.max_list = max_list
        self.base_list = []
        for i in range(len(tensors)):
            b = 1
            for m in max_list[i + 1:]:
                b *= int(m)
            self.base_list.append(b)
        data = torch.zeros_like(tensors[0], dtype=torch.int64)
        for a, b in zip(tensors, self.base_list):
            data = data + a.long() * b
        self.data = data

    @property
    def shape(self):
        return self.data.shape

    @property
    def device(self):
        return self.data.device

    def restore(self, data=None):
        """Split (possibly sorted / filtered) composite keys back into the input columns."""
        composite = self.data if data is None else data
        out = []
        for b, dt in zip(self.base_list, self.dtype_list):
            out.append(torch.div(composite, b, rounding_mode='floor').type(dt) if b != 0
                       else torch.zeros_like(composite).type(dt))
            composite = composite % b if b != 0 else composite
        return out

    def __repr__(self):
        return f"{self.__class__.__name__}(shape={self.shape}, dtype={self.dtype}, device={self.device})"


def _workspace(lib, n, device):
    nbytes = lib.dva_lex_workspace_bytes(n)
    if nbytes < 0:
        check(int(nbytes), "dva_lex_workspace_bytes")
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device), int(nbytes)


def argsort_keys(keys):
    """Stable argsort of an int64 key tensor on the device -> (order, sorted_keys)."""
    lib = _lib.load()
    require_device(keys)
    keys = keys.contiguous()
    n = keys.shape[0]
    order = torch.empty(n, dtype=torch.int64, device=keys.device)
    keys_sorted = torch.empty_like(keys)
    ws, nbytes = _workspace(lib, n, keys.device)
    check(lib.dva_argsort_i64(ptr(keys), n, ptr(order), ptr(keys_sorted), ptr(ws), nbytes, stream_of(keys)),
          "dva_argsort_i64")
    return order, keys_sorted


def argunique_keys(keys):
    """Index of the first occurrence of every distinct key, in key order."""
    lib = _lib.load()
    require_device(keys)
    keys = keys.contiguous()
    n = keys.shape[0]
    first = torch.empty(n, dtype=torch.int64, device=keys.device)
    n_unique = torch.zeros(1, dtype=torch.int64, device=keys.device)
    ws, nbytes = _workspace(lib, n, keys.device)
    check(lib.dva_argunique_i64(ptr(keys), n, ptr(first), ptr(n_unique), ptr(ws), nbytes, stream_of(keys)),
          "dva_argunique_i64")
    return first[:int(n_unique.item())]


def lexargsort(*args, use_cuda=True):
    """Indices sorting the input tensors in lexicographic order (reference :51-62). Stable."""
    return argsort_keys(CompositeTensor(*args).data)[0]


def lexsort(*args, use_cuda=True):
    """Input tensors sorted in lexicographic order (reference :36-48)."""
    comp = CompositeTensor(*args)
    out = comp.restore(argsort_keys(comp.data)[1])
    return out if len(out) > 1 else out[0]


def lexargunique(*args, use_cuda=True):
    """Indices of the first occurrence of each unique row, in lexicographic order (reference :80-94)."""
    return argunique_keys(CompositeTensor(*args).data)


def lexunique(*args, use_cuda=True):
    """Unique rows of the input tensors in lexicographic order (reference :65-77)."""
    comp = CompositeTensor(*args)
    out = comp.restore(comp.data[argunique_keys(comp.data)])
    return out if len(out) > 1 else out[0]
