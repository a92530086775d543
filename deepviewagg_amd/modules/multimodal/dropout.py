"""Hard modality dropout (reference behaviour: torch_points3d/modules/multimodal/dropout.py:5-15).

Training: one Bernoulli draw per call decides whether the WHOLE modality tensor survives (kept as is) or is
zeroed; evaluation: the tensor is scaled by 1 / (1 - p).  The draw is made on the tensor's device, so a branch
never synchronises with the host.
"""
import torch
from torch import nn


class ModalityDropout(nn.Module):
    def __init__(self, p=0, inplace=False, **kwargs):
        # ``inplace`` is accepted because UnimodalBranch passes it (modules.py:276-280); results are never in place
        super().__init__()
        if not 0 <= p <= 1:
            raise AssertionError('p must be in [0, 1].')
        self.p = p

    def extra_repr(self):
        return f"p={self.p}"

    def forward(self, x):
        if self.training:
            gate = torch.rand((), device=x.device) > self.p     # 0-dim bool tensor
            return x * gate.to(x.dtype)
        return x * (1.0 / (1.0 - self.p))
