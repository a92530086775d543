"""Bimodal fusion (reference: torch_points3d/modules/multimodal/fusion.py:7-53)."""
import torch
import torch.nn as nn


class BimodalFusion(nn.Module):
    """Fuse the pooled modality features [N, C_2] into the main 3D features [N, C_1].

    modes: 'residual' (a + b), 'concatenation' (cat(a, b)), 'both' (cat(a, a + b)),
    'modality' (b).  Either input may be None, in which case the other one is returned.
    """

    MODES = ['residual', 'concatenation', 'both', 'modality']

    def __init__(self, mode='residual', **kwargs):
        super().__init__()
        if mode not in self.MODES:
            raise NotImplementedError(
                f"Unknown fusion mode='{mode}'. Please choose among supported modes: {self.MODES}.")
        self.mode = mode

    def forward(self, x_main, x_mod):
        if x_main is None:
            return x_mod
        if x_mod is None:
            return x_main
        # sparse tensors carry their features in .F
        if not isinstance(x_mod, torch.Tensor):
            x_mod = x_mod.F
        if self.mode == 'residual':
            return x_main + x_mod
        if self.mode == 'concatenation':
            return torch.cat((x_main, x_mod), dim=-1)
        if self.mode == 'both':
            return torch.cat((x_main, x_main + x_mod), dim=-1)
        return x_mod

    def extra_repr(self) -> str:
        return f"mode={self.mode}"
