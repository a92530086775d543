"""Bimodal fusion (reference: torch_points3d/modules/multimodal/fusion.py:7-53)."""
import torch
import torch.nn as nn


def _fusable(x_main, x_mod):
    """Device tensors the one-pass concat + promote kernel covers (fp32 main, fp32 / bf16 modality, [N, 4k])."""
    return (x_main.is_cuda and x_mod.is_cuda and x_main.dim() == 2 and x_mod.dim() == 2
            and x_main.dtype == torch.float32 and x_mod.dtype in (torch.float32, torch.bfloat16)
            and x_main.shape[0] == x_mod.shape[0] and x_main.shape[1] % 4 == 0 and x_mod.shape[1] % 4 == 0
            and x_main.shape[1] + x_mod.shape[1] <= 1024)


class _ConcatCast(torch.autograd.Function):
    """torch.cat((x_main, x_mod), -1) with the promotion to fp32 in the same pass (dva_concat_cast_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x_main, x_mod):
        from ... import _lib, ops
        from ..._lib import check, ptr, stream_of
        lib = _lib.load()
        x_main, x_mod = x_main.contiguous(), x_mod.contiguous()
        N, Ca, Cb = x_main.shape[0], x_main.shape[1], x_mod.shape[1]
        out = torch.empty((N, Ca + Cb), dtype=torch.float32, device=x_main.device)
        with ops._timed("fusion_concat", N * (Ca * 4 + Cb * x_mod.element_size() + (Ca + Cb) * 4)):
            check(lib.dva_concat_cast_fwd(ptr(x_main), ptr(x_mod), ptr(out), N, Ca, Cb, ops.dtype_code(x_mod),
                                          stream_of(out)), "dva_concat_cast_fwd")
        ctx.meta = (N, Ca, Cb, x_mod.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        from ... import _lib, ops
        from ..._lib import check, ptr, stream_of
        lib = _lib.load()
        N, Ca, Cb, dt = ctx.meta
        gout = gout.contiguous().float()
        ga = torch.empty((N, Ca), dtype=torch.float32, device=gout.device)
        gb = torch.empty((N, Cb), dtype=dt, device=gout.device)
        with ops._timed("fusion_concat_bwd", N * (Ca * 4 + Cb * gb.element_size() + (Ca + Cb) * 4)):
            check(lib.dva_concat_cast_bwd(ptr(gout), ptr(ga), ptr(gb), N, Ca, Cb, ops.dtype_code(gb), stream_of(gout)),
                  "dva_concat_cast_bwd")
        return ga, gb


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
            if _fusable(x_main, x_mod):
                return _ConcatCast.apply(x_main, x_mod)
            return torch.cat((x_main, x_mod), dim=-1)
        if self.mode == 'both':
            return torch.cat((x_main, x_main + x_mod), dim=-1)
        return x_mod

    def extra_repr(self) -> str:
        return f"mode={self.mode}"
