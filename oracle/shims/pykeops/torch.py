"""Stand-in for pykeops.torch (v1.4.2 is the reference's pin, install.sh:141; absent from this image).
TEST INFRASTRUCTURE ONLY (oracle/gen_golden.py).

Only what the reference's NeighborhoodBasedMappingFeatures writes is provided
(core/data_transform/multimodal/image.py:499-508):

    d = ((LazyTensor(x[:, None, :]) - LazyTensor(x[None, :, :])) ** 2).sum(dim=2);  d.argKmin(K, dim=1)

evaluated densely, row block by row block, with torch broadcasting in the dtype of the operands (fp32: the squared
distance is ((dx^2 + dy^2) + dz^2), left to right).  `argKmin` = the indices of the K smallest entries along `dim`;
KeOps does not specify the order among equal distances: here ties go to the lower index (stable sort)."""
import torch


class LazyTensor:
    def __init__(self, x):
        self.x = x if isinstance(x, torch.Tensor) else torch.as_tensor(x)

    def __sub__(self, other):
        return _Expr(lambda lo, hi: _rows(self.x, lo, hi) - _rows(other.x, lo, hi), _n_rows(self.x, other.x))


def _n_rows(a, b):
    return max(a.shape[0], b.shape[0])


def _rows(x, lo, hi):
    return x if x.shape[0] == 1 else x[lo:hi]


class _Expr:
    """f(lo, hi) -> dense block [hi - lo, M, D] (or [hi - lo, M] after the reduction over the last axis)."""

    def __init__(self, f, n):
        self.f, self.n = f, n

    def __pow__(self, p):
        return _Expr(lambda lo, hi: self.f(lo, hi) ** p, self.n)

    def sum(self, dim):
        assert dim == 2, "the shim reduces the coordinate axis only"

        def f(lo, hi):
            e = self.f(lo, hi)
            acc = e[..., 0]
            for c in range(1, e.shape[-1]):       # left to right, like a generated reduction
                acc = acc + e[..., c]
            return acc
        return _Expr(f, self.n)

    def argKmin(self, K, dim):
        assert dim == 1
        out = []
        for lo in range(0, self.n, 256):
            d = self.f(lo, min(lo + 256, self.n))
            out.append(torch.sort(d, dim=1, stable=True)[1][:, :K])
        return torch.cat(out, 0)
