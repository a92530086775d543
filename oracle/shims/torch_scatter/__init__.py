"""Test-infrastructure stub (NOT product code): torch-native stand-in for the
`torch_scatter` wheel (absent from the image), used only by oracle/gen_golden.py
to run the reference's Python source. Semantics pinned by the reference's own
statements: empty CSR groups reduce to 0 (pooling.py:870), CSR reduction is
deterministic (pooling.py:885-887), scatter_min/max arg of an empty group is
src.size(dim) (pooling.py:136), max/min ties go to the first element."""
import torch


def _expand_ptr(indptr, n_rows):
    sizes = indptr[1:] - indptr[:-1]
    return torch.arange(sizes.shape[0], device=indptr.device).repeat_interleave(sizes)


def segment_csr(src, indptr, out=None, reduce='sum'):
    assert indptr.dim() == 1
    n = indptr.shape[0] - 1
    sizes = (indptr[1:] - indptr[:-1])
    idx = _expand_ptr(indptr, src.shape[0])
    shape = (n,) + tuple(src.shape[1:])
    if reduce in ('sum', 'add', 'mean'):
        res = torch.zeros(shape, dtype=src.dtype, device=src.device)
        res = res.index_add(0, idx, src)
        if reduce == 'mean':
            cnt = sizes.clamp(min=1).to(src.dtype)
            res = res / cnt.view((-1,) + (1,) * (src.dim() - 1))
        return res
    if reduce in ('max', 'min'):
        # first-occurrence arg semantics, gradient routed to the arg only
        big = float('-inf') if reduce == 'max' else float('inf')
        full = torch.full(shape, big, dtype=src.dtype, device=src.device)
        idx_e = idx.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
        red = 'amax' if reduce == 'max' else 'amin'
        val = full.scatter_reduce(0, idx_e, src.detach(), red, include_self=True)
        # arg = first row in the group equal to the extremum
        rows = torch.arange(src.shape[0], device=src.device)
        rows = rows.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
        is_ext = src.detach() == val.gather(0, idx_e) if src.shape[0] > 0 else src.detach() == src.detach()
        cand = torch.where(is_ext, rows, torch.full_like(rows, src.shape[0]))
        arg = torch.full(shape, src.shape[0], dtype=torch.long, device=src.device)
        arg = arg.scatter_reduce(0, idx_e, cand, 'amin', include_self=True)
        empty = (sizes == 0).view((-1,) + (1,) * (src.dim() - 1)).expand(shape)
        src0 = torch.cat([src, torch.zeros((1,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)])
        cols = None
        res = src0.gather(0, arg) if src.dim() > 1 else src0[arg]
        res = torch.where(empty, torch.zeros_like(res), res)
        return res.clone()
    raise ValueError(reduce)


def scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = (dim_size,) + tuple(src.shape[1:])
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    if reduce in ('sum', 'add'):
        return res.index_add(0, index, src)
    if reduce == 'mean':
        res = res.index_add(0, index, src)
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device).index_add(
            0, index, torch.ones_like(index, dtype=src.dtype)).clamp(min=1)
        return res / cnt.view((-1,) + (1,) * (src.dim() - 1))
    raise ValueError(reduce)


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, 'sum')


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, 'mean')


def _scatter_ext(src, index, dim, dim_size, mode):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = (dim_size,) + tuple(src.shape[1:])
    idx_e = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    big = float('-inf') if mode == 'max' else float('inf')
    if not src.is_floating_point():
        info = torch.iinfo(src.dtype)
        big = info.min if mode == 'max' else info.max
    full = torch.full(shape, big, dtype=src.dtype, device=src.device)
    val = full.scatter_reduce(0, idx_e, src, 'amax' if mode == 'max' else 'amin', include_self=True)
    rows = torch.arange(src.shape[0], device=src.device)
    rows = rows.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    is_ext = src == val.gather(0, idx_e)
    cand = torch.where(is_ext, rows, torch.full_like(rows, src.shape[0]))
    arg = torch.full(shape, src.shape[0], dtype=torch.long, device=src.device)
    arg = arg.scatter_reduce(0, idx_e, cand, 'amin', include_self=True)
    empty = arg == src.shape[0]
    val = torch.where(empty, torch.zeros_like(val), val)
    return val, arg


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    return _scatter_ext(src, index, dim, dim_size, 'max')


def scatter_min(src, index, dim=0, out=None, dim_size=None):
    return _scatter_ext(src, index, dim, dim_size, 'min')
