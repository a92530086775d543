"""Hard modality dropout (reference: torch_points3d/modules/multimodal/dropout.py:5-15)."""
import torch
import torch.nn as nn


class ModalityDropout(nn.Module):
    """Drop the WHOLE modality with probability p at train time; scale by 1/(1-p) at eval time.

    ``inplace`` is accepted (UnimodalBranch passes it, modules.py:276-280) and ignored.
    """

    def __init__(self, p=0, inplace=False, **kwargs):
        super().__init__()
        assert 0 <= p <= 1, 'p must be in [0, 1].'
        self.p = p

    def forward(self, x):
        if not self.training:
            return x.mul(1 / (1 - self.p))
        keep = bool(torch.rand(1).item() > self.p)
        return x.mul(float(keep))
