from .base_modules import MLP, FastBatchNorm1d, Identity, BaseModule  # noqa: F401
