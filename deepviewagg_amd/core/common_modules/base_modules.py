"""Building blocks the pooling modules are made of (reference:
torch_points3d/core/common_modules/base_modules.py:8-48, :131-156).

Only the module tree / parameter names are part of the drop-in contract (state-dict
keys such as ``E_map.mlp_elt_1.0.0.weight`` and ``...0.1.batch_norm.running_mean`` must load the
published checkpoints); the maths is a Linear -> BatchNorm1d -> LeakyReLU(0.2) stack.
"""
import torch
from torch import nn
import torch.nn.functional as F


class BaseModule(nn.Module):
    @property
    def nb_params(self):
        """Number of trainable parameters (base_modules.py:13-21)."""
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class Identity(BaseModule):
    def forward(self, data):
        return data


class FastBatchNorm1d(BaseModule):
    """BatchNorm over the rows of a [rows, C] matrix (or the (B, N) dims of [B, N, C]).

    The reference reshapes to [1, C, rows] so that nn.BatchNorm1d treats rows as the length axis
    (base_modules.py:139-148); statistics are identical to normalising dim 0 directly, which is
    what is done here.  The wrapped ``batch_norm`` keeps the reference's parameter names.
    """

    def __init__(self, num_features, momentum=0.1, **kwargs):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(num_features, momentum=momentum, **kwargs)

    def forward(self, x):
        if x.dim() == 2:
            return self.batch_norm(x)
        if x.dim() == 3:
            return self.batch_norm(x.permute(0, 2, 1)).permute(0, 2, 1)
        raise ValueError("Non supported number of dimensions {}".format(x.dim()))


def MLP(channels, activation=None, bn_momentum=0.1, bias=True):
    """Sequential of [Linear, FastBatchNorm1d, activation] blocks (base_modules.py:38-48).

    The reference shares one LeakyReLU(0.2, inplace=True) instance across every block as a default
    argument; the activation is stateless so a fresh instance per MLP is equivalent.
    """
    if activation is None:
        activation = nn.LeakyReLU(0.2, inplace=True)
    return nn.Sequential(*[
        nn.Sequential(
            nn.Linear(channels[i - 1], channels[i], bias=bias),
            FastBatchNorm1d(channels[i], momentum=bn_momentum),
            activation)
        for i in range(1, len(channels))])
