"""Bimodal CSR pooling modules backed by the gfx950 kernels.

Drop-in mirror of ``torch_points3d/modules/multimodal/pooling.py`` (reference): same class names,
constructor keywords, ``forward(x_main, x_mod, x_map, csr_idx)`` contract, parameter/state-dict
names and ``save_last`` attributes, so that ``ModalityFactory.get_module``
(models/base_architectures/unet.py:69-101) resolves YAML ``module_name`` entries to these classes.

What runs where:
  * CSR reductions, segment softmax, attention-weighted sum and gating -> HIP kernels
    (``deepviewagg_amd.ops``), hand-written forward AND backward;
  * the small dense layers (Linear / BatchNorm1d / LeakyReLU of ``MLP``) -> PyTorch-ROCm
    (rocBLAS/hipBLASLt GEMMs), as the north star prescribes for dense stages.

The order of 3D points in the main modality must match the order of the groups in ``csr_idx``
(reference docstring, pooling.py:30-33).
"""
import math
import sys

import torch
import torch.nn as nn

from ...core.common_modules import MLP
from ... import ops
from ... import fused_deepset, fused_chain, fused_chain_f32, fused_bilinear
from ...ops import (segment_csr, gather_csr, segment_gather_csr,  # noqa: F401 (re-exported)
                    segment_softmax_csr)

_local_modules = sys.modules[__name__]


def _dense_index(csr_idx, device):
    """Group id of every row covered by ``csr_idx`` (the reference's ``_last_idx``)."""
    sizes = csr_idx[1:] - csr_idx[:-1]
    return torch.arange(csr_idx.shape[0] - 1, device=device).repeat_interleave(sizes), sizes


def _materialize(x_mod):
    return x_mod.materialize() if isinstance(x_mod, ops.LAZY_TYPES) else x_mod


def batchnorm_act_rows(y, bn, slope, counts=None, n=None):
    """``leaky_slope(BatchNorm1d(y))`` on the rows of ``y`` [R, C] with the HIP row kernels (slope 0 = ReLU,
    slope 1 = no activation).  ``counts`` weights the batch statistics (row r stands for counts[r] gathered
    rows, ``n`` of them in total); running statistics are updated as ``nn.BatchNorm1d`` does."""
    batch_stats = bn.training or not bn.track_running_stats
    # the one-launch path hands raw fp32 pointers of the module's parameters / buffers to the kernels: anything else
    # (a module cast with .half() / .bfloat16()) takes the generic branch below
    f32 = all(t is None or t.dtype == torch.float32
              for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
    if bn.momentum is not None and y.is_cuda and f32:
        # statistics -> constants (+ running statistics) in one launch
        n = (float(y.shape[0]) if n is None else float(n)) if batch_stats else 1.0
        sums = ops.rowbn_sums(y, counts) if batch_stats else None
        tab = ops.bn_table(sums, n, bn, batch_stats)
        return ops.rowbn_act(y, counts, bn.weight if bn.affine else None, bn.bias if bn.affine else None,
                             None, None, n, batch_stats, slope, bn_tab=tab)
    if batch_stats:
        n = float(y.shape[0]) if n is None else float(n)
        s1, s2 = ops.rowbn_stats(y, counts)
        mean = s1 / n
        var = (s2 / n - mean * mean).clamp_(min=0.0)
        if bn.training and bn.track_running_stats:
            with torch.no_grad():
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                bn.running_mean.mul_(1 - m).add_(m * mean.float())
                bn.running_var.mul_(1 - m).add_(m * (var * (n / max(n - 1.0, 1.0))).float())
                bn.num_batches_tracked += 1
        mean, var = mean.float(), var.float()
    else:
        n, mean, var = 1.0, bn.running_mean, bn.running_var
    invstd = torch.rsqrt(var + bn.eps)
    return ops.rowbn_act(y, counts, bn.weight if bn.affine else None, bn.bias if bn.affine else None,
                         mean, invstd, n, batch_stats, slope)


def mlp_on_gathered_rows(mlp, rows, counts, n_views=None, first_linear_done=False):
    """Evaluate ``mlp(rows[row_idx])`` WITHOUT gathering: returns ``out_rows`` such that
    ``out_rows[row_idx] == mlp(rows[row_idx])`` row for row.

    ``mlp`` is a Sequential of [Linear, FastBatchNorm1d, activation] blocks (MLP()).  Linear and the
    activation act on each row independently, and the train-mode BatchNorm statistics over the P
    gathered rows equal the statistics over the R map rows weighted by ``counts`` (how many atoms
    gather each row).  P/R is ~128 on the headline workload, so the dense layers cost 1/128 of the
    per-view evaluation of the reference (pooling.py:245,275) and no [P, C] tensor is materialised.
    Backward is plain autograd over the [R, C] tensors.
    ``first_linear_done``: ``rows`` already are the output of the first block's Linear (hoisted by the caller).
    """
    x = rows
    n = None
    for i, block in enumerate(mlp):
        lin, bn, act = block[0], block[1].batch_norm, block[2]
        slope = _leaky_slope(act)
        if slope is None or x.dtype not in (torch.float32, torch.bfloat16):
            if first_linear_done:
                raise NotImplementedError("hoisted first Linear with an activation the row kernels do not cover")
            return _mlp_on_gathered_rows_torch(mlp, rows, counts)
        y = x if (first_linear_done and i == 0) else ops.tall_linear(x, lin.weight, lin.bias)
        if n is None and (bn.training or not bn.track_running_stats):
            # number of gathered rows (views); callers pass it to avoid a device synchronisation
            n = float(n_views if n_views is not None else (counts.sum() if counts is not None else x.shape[0]))
        x = batchnorm_act_rows(y, bn, slope, counts, n)
    return x


def _mlp_rows(mlp, x):
    """``mlp(x)`` for a materialised [V, C] tensor through the row kernels (every row counts once): the library
    BatchNorm path is two orders of magnitude slower at V ~ 3e7 rows (64-bit indexing)."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] > 0:
        return mlp_on_gathered_rows(mlp, x, None, x.shape[0])
    return mlp(x)


def _hoisted_first_linear(mlp, x_mod):
    """E_mod on a lazily BILINEAR-gathered feature map outside the fused path (C_out > 64, fp32 maps, ...): the first
    Linear (no bias) commutes with the interpolation, so it runs on the R map rows and the gather carries C_out
    channels instead of C_in -- the per-view GEMM V x C_in x C_out of the reference (pooling.py:245,275 on the output
    of image.py:105-170) and the [V, C_in] tensor disappear (KITTI-360 pyramid: 256 -> 128, 512 -> 256).  Returns
    (E_mod(x_mod) as a [V, C_out] tensor, True), or (the materialised [V, C_in] gather, False) when it does not apply."""
    lin, bn, act = mlp[0][0], mlp[0][1].batch_norm, mlp[0][2]
    rows = x_mod.rows
    ok = (x_mod.exact and lin.bias is None and rows.is_cuda and rows.dtype in (torch.float32, torch.bfloat16)
          and all(_leaky_slope(b[2]) is not None for b in mlp))
    if not ok:
        return x_mod.materialize(), False
    y_rows = ops.tall_linear(rows, lin.weight)                 # [R, C_out]
    z_a = x_mod.materialize(rows=y_rows)                        # [V, C_out]: interp(x) W^T = interp(x W^T)
    return mlp_on_gathered_rows(mlp, z_a, None, z_a.shape[0], first_linear_done=True), True


def _leaky_slope(act):
    """Negative slope of the activations the fused row kernels cover, else None."""
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    if isinstance(act, nn.ReLU):
        return 0.0
    return None


def _mlp_on_gathered_rows_torch(mlp, rows, counts):
    """Generic composition of ``mlp_on_gathered_rows`` for activations without a fused kernel."""
    if counts is None:
        return mlp(rows)
    w = counts.to(torch.float32).unsqueeze(1)
    n = w.sum()
    x = rows
    for block in mlp:
        lin, bn, act = block[0], block[1].batch_norm, block[2]
        y = lin(x)
        yf = y.float()
        if bn.training or not bn.track_running_stats:
            mean = (w * yf).sum(0) / n
            var = (w * (yf - mean) ** 2).sum(0) / n
            if bn.training and bn.track_running_stats:
                with torch.no_grad():
                    m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - m).add_(m * mean)
                    bn.running_var.mul_(1 - m).add_(m * var * n / (n - 1))
                    bn.num_batches_tracked += 1
        else:
            mean, var = bn.running_mean, bn.running_var
        z = (yf - mean) * torch.rsqrt(var + bn.eps)
        if bn.affine:
            z = z * bn.weight + bn.bias
        x = act(z.to(y.dtype))
    return x


class _SaveLast:
    """Shared bookkeeping of the optional ``save_last`` debugging outputs."""

    def _init_save_last(self, save_last, extra=()):
        self.save_last = save_last
        for k in ('x_map', 'x_mod', 'idx', 'view_num') + tuple(extra):
            setattr(self, f'_last_{k}', None)

    def _save(self, x_map, x_mod, csr_idx, **extra):
        if not self.save_last:
            return
        self._last_x_map = x_map
        self._last_x_mod = _materialize(x_mod)
        self._last_idx, self._last_view_num = _dense_index(csr_idx, x_mod.device)
        for k, v in extra.items():
            setattr(self, f'_last_{k}', v)


class BimodalCSRPool(nn.Module, _SaveLast):
    """max / mean / min / sum pooling of modality features over CSR groups
    (reference pooling.py:14-71). Used for atomic-level (pixels -> view) and view-level
    (views -> point) aggregation; empty groups receive zeros."""

    _POOLING_MODES = ['max', 'mean', 'min', 'sum']

    def __init__(self, mode='max', save_last=False, **kwargs):
        super().__init__()
        assert mode in self._POOLING_MODES, \
            f"Unsupported mode '{mode}'. Expected one of: {self._POOLING_MODES}"
        self._mode = mode
        self._init_save_last(save_last)

    def forward(self, x_main, x_mod, x_map, csr_idx):
        """x_main [N, F_main] (unused), x_mod [V, F_mod], x_map [V, F_map] (unused), csr_idx [N+1]."""
        if isinstance(x_mod, ops.LAZY_TYPES):
            if x_map is None and x_mod.exact and csr_idx.shape[0] - 1 == x_mod.shape[0]:
                # ATOMIC pooling (UnimodalBranch passes x_map=None there) of an exact mapping (one pixel
                # per view, atomic CSR = arange): every group holds exactly one row, so max / min / mean
                # / sum are all the identity -> stay lazy.  At the view level (x_map given) N == V does
                # not imply one view per point (csr = [0, 2, 2, 3]): always reduce there.
                self._save(x_map, x_mod, csr_idx)
                return x_mod
            if (x_map is None and self._mode == 'max' and not self.save_last
                    and csr_idx.shape[0] - 1 < x_mod.shape[0] and ops.gather_segment_max_applicable(x_mod, csr_idx)):
                # ATOMIC max pool of a NON-exact mapping (several pixels per view, round 4): gather and max in one kernel,
                # no [P, C] tensor; the result stays lazy at the VIEW level (identity gather over the pooled [V, C] rows) so
                # that the view pooling keeps its fused path
                return ops.gather_segment_max(x_mod, csr_idx)
            if (x_map is not None and self._mode == 'max' and not self.save_last
                    and ops.gather_segment_max_applicable(x_mod, csr_idx)):
                # VIEW-level max pool of lazily gathered values: the same fused kernel, a plain [N, C] tensor out
                return ops.gather_segment_max(x_mod, csr_idx).rows
            x_mod = x_mod.materialize()
        x_pool = segment_csr(x_mod, csr_idx, reduce=self._mode)
        self._save(x_map, x_mod, csr_idx)
        return x_pool

    def extra_repr(self) -> str:
        return f'mode={self._mode}, save_last={self.save_last}'


class HeuristicBimodalCSRPool(nn.Module, _SaveLast):
    """Select, for each group, the row whose chosen mapping feature is max / min
    (reference pooling.py:74-156). Unseen groups receive zeros."""

    _MODES = ['max', 'min']
    _FEATURES = [
        'normalized_depth', 'linearity', 'planarity', 'scattering',
        'orientation_to_the_surface', 'normalized_pixel_height', 'density', 'occlusion']

    def __init__(self, mode='max', feat=0, save_last=False, **kwargs):
        super().__init__()
        assert mode in self._MODES, f"Unsupported mode '{mode}'. Expected one of: {self._MODES}."
        self._mode = mode
        feat = self._FEATURES.index(feat) if isinstance(feat, str) else feat
        assert feat < len(self._FEATURES), \
            f"Feat={feat} is too large. Expected feat<{len(self._FEATURES)}."
        self._feat = feat
        self._init_save_last(save_last)

    def forward(self, x_main, x_mod, x_map, csr_idx):
        if isinstance(x_mod, ops.GatheredFeatures) and not self.save_last:
            # lazily gathered values (round 4): only the SELECTED view of every point is gathered -- N rows of the map
            # instead of the [V, C] tensor (x_pool[i] = rows[row_idx[arg_i]], zeros for unseen points)
            _, arg_idx = ops.segment_csr_arg(x_map[:, self._feat].float(), csr_idx, reduce=self._mode)
            arg_idx = arg_idx.reshape(-1).long()
            seen = arg_idx >= 0
            sel = x_mod.row_idx.long()[arg_idx.clamp(min=0)] if x_mod.row_idx.shape[0] else arg_idx.clamp(min=0)
            x_pool = ops.gather_rows(x_mod.rows, sel.to(torch.int32)) if x_mod.row_idx.shape[0] else \
                x_mod.rows.new_zeros((arg_idx.shape[0], x_mod.rows.shape[1]))
            # unseen points: exact zeros whatever the (unrelated) gathered row holds -- 0 * Inf would be NaN
            return torch.where(seen.unsqueeze(1), x_pool, torch.zeros((), dtype=x_pool.dtype, device=x_pool.device))
        x_mod = _materialize(x_mod)
        # arg of the per-group extremum of the heuristic feature (first row on ties, -1 if unseen)
        _, arg_idx = ops.segment_csr_arg(x_map[:, self._feat].float(), csr_idx, reduce=self._mode)
        arg_idx = arg_idx.reshape(-1).long()
        # unseen points index an appended zero row (pooling.py:140-143)
        arg_idx = torch.where(arg_idx < 0, torch.full_like(arg_idx, x_mod.shape[0]), arg_idx)
        x_mod_0 = torch.cat((x_mod, torch.zeros_like(x_mod[[0]])))
        x_pool = x_mod_0[arg_idx]
        self._save(x_map, x_mod, csr_idx)
        return x_pool

    def extra_repr(self) -> str:
        return f'mode={self._mode}, feat={self._FEATURES[self._feat]}, save_last={self.save_last}'


class Gating(nn.Module):
    """Rectified-tanh gating with learnable affine correction: tanh(relu(w * x + b))
    (reference pooling.py:690-715). Inside the pooling modules the gate is evaluated by the fused
    view-attention kernel from ``weight`` / ``bias``; ``forward`` is the stand-alone form."""

    def __init__(self, num_groups, weight=True, bias=True, activation='tanh+'):
        super().__init__()
        self.num_groups = num_groups
        self.weight = nn.Parameter(torch.ones(1, num_groups)) if weight else None
        self.bias = nn.Parameter(torch.zeros(1, num_groups)) if bias else None
        if activation not in ('tanh+', 'sigmoid'):
            raise ValueError(f"Activation '{activation}' not supported for Gating")
        self._activation = activation

    def forward(self, x):
        if self.weight is not None:
            x = x * self.weight
        if self.bias is not None:
            x = x + self.bias
        # NB: like the reference (pooling.py:710-711), forward always applies tanh(relu(.))
        return torch.tanh(torch.relu(x)).view(-1, self.num_groups).squeeze(1)

    def extra_repr(self) -> str:
        return f'num_groups={self.num_groups}, weight={self.weight is not None}, ' \
               f'bias={self.bias is not None}'


def _pool_with_attention(module, x_mod, compatibilities, csr_idx):
    """softmax -> attention-weighted sum -> gating, one fused kernel (pooling.py:284-300)."""
    G = module.G
    kw = dict(gate_w=G.weight if G is not None else None, gate_b=G.bias if G is not None else None,
              scaling=module.group_scaling)
    if isinstance(x_mod, ops.GatheredFeatures):
        x_pool, attentions, gating = ops.view_gather_attention(
            x_mod.rows, x_mod.row_idx, compatibilities, csr_idx, plan=x_mod.plan, **kw)
    else:
        x_pool, attentions, gating = ops.view_attention(x_mod, compatibilities, csr_idx, **kw)
    if G is not None and module.num_groups == 1:
        gating = gating.squeeze(1)
    return x_pool, attentions, (gating if G is not None else None)


class GroupBimodalCSRPool(nn.Module, _SaveLast):
    """DeepViewAgg view pooling: attention over the views of each point, scores computed from the
    mapping features only (optionally also from the modality features), one score per channel
    group, optional gating (reference pooling.py:159-319).

    Example (pooling.py:185-204)::

        csr_idx = torch.LongTensor([0, 4, 4, 5, 10, 20]).cuda()
        module = GroupBimodalCSRPool(in_map=3, in_mod=7, num_groups=2).cuda()
        module(None, torch.rand(20, 7).cuda(), torch.rand(20, 3).cuda(), csr_idx)   # [5, 7]
    """

    def __init__(
            self, in_map=None, in_mod=None, out_mod=None, num_groups=1,
            use_mod=False, gating=True, group_scaling=True, save_last=False,
            nc_inner=32, map_encoder='DeepSetFeat', **kwargs):
        super().__init__()
        self.nc_inner = nc_inner
        self._init_save_last(save_last, extra=('C', 'A', 'G'))

        assert 1 <= num_groups <= in_mod, \
            f"Number of groups must be between 1 and in_mod={in_mod}."
        out_mod = in_mod if out_mod is None else out_mod
        self.in_mod = in_mod
        self.out_mod = out_mod
        self.use_mod = use_mod
        self.num_groups = num_groups
        self.group_scaling = group_scaling

        # E_map: mapping features -> nc_inner;  E_mod: modality features -> values
        self.E_map = getattr(_local_modules, map_encoder)(in_map, nc_inner, **kwargs)
        self.E_mod = MLP([in_mod, out_mod, out_mod], bias=False)
        if self.use_mod:
            in_mix, out_mix = nc_inner + out_mod, nc_inner
            mid_mix = nearest_power_of_2((in_mix + out_mix) / 2, out_mix * 2)
            self.E_mix = MLP([in_mix, mid_mix, out_mix], bias=False)
        self.E_score = nn.Linear(nc_inner, num_groups, bias=True)
        self.G = Gating(num_groups, bias=True) if gating else None

    def forward(self, x_main, x_mod, x_map, csr_idx):
        """x_main [N, F_main] (unused), x_mod [V, F_mod], x_map [V, F_map], csr_idx [N+1] -> [N, out_mod]."""
        val_rows = None
        emod_done = False
        if isinstance(x_mod, ops.InterpolatedFeatures):
            # bilinear gather (interpolate=True): E_mod per view inside the chain kernels, its first Linear on the map
            # rows (fused_bilinear.py); anything the fused path does not cover materialises the reference's [V, C]
            if fused_bilinear.applicable(self, x_mod, x_map, csr_idx):
                return fused_bilinear.pool(self, x_mod, x_map, csr_idx)
            x_mod, emod_done = _hoisted_first_linear(self.E_mod, x_mod)
        if isinstance(x_mod, ops.GatheredFeatures) and fused_chain.applicable(self, x_mod, x_map, csr_idx):
            # bf16 recompute chain: E_mod on the map rows, then ONE view kernel (DeepSetFeat scores, softmax,
            # row gather, weighted sum, gate) -- no [V, .] activation tensor at all (fused_chain.py)
            val_rows = mlp_on_gathered_rows(self.E_mod, x_mod.rows, x_mod.counts, x_mod.shape[0])
            if val_rows.dtype == torch.bfloat16:
                return fused_chain.chain_pool(self, x_mod.with_rows(val_rows), x_map, csr_idx)
        fused_scores = (not self.use_mod and not self.save_last
                        and fused_deepset.applicable(self.E_map, self.E_score, x_map))
        if fused_scores:
            # DeepSetFeat + E_score in the fused row-streaming kernels (fp32, hand-written backward)
            if fused_chain_f32.applicable(self.E_map, self.E_score, x_map, csr_idx):
                # fp32 chain on the fp32 matrix cores (csrc/chain_f32.hip): two stored [V, 32] tensors instead of thirteen
                compatibilities = fused_chain_f32.chain_scores(self.E_map, self.E_score, x_map, csr_idx)
            else:
                compatibilities = fused_deepset.deepset_linear(self.E_map, self.E_score, x_map, csr_idx)
        else:
            x_map = self.E_map(x_map, csr_idx)
        if isinstance(x_mod, ops.GatheredFeatures) and not self.use_mod:
            # lazy nearest gather: E_mod runs on the map rows, the gather is fused into the
            # attention kernel (no [V, C] tensor exists on this path)
            if val_rows is None:
                val_rows = mlp_on_gathered_rows(self.E_mod, x_mod.rows, x_mod.counts, x_mod.shape[0])
            if not fused_scores:
                compatibilities = self.E_score(x_map)
            x_mod = x_mod.with_rows(val_rows)
        else:
            if not emod_done:
                x_mod = _mlp_rows(self.E_mod, _materialize(x_mod))
            if self.use_mod:
                compatibilities = self.E_score(self.E_mix(torch.cat([x_map, x_mod.to(x_map.dtype)], dim=1)))
            elif not fused_scores:
                compatibilities = self.E_score(x_map)
        x_pool, attentions, gating = _pool_with_attention(self, x_mod, compatibilities, csr_idx)
        self._save(x_map, x_mod, csr_idx, C=compatibilities, A=attentions,
                   **({'G': gating} if self.G is not None else {}))
        return x_pool

    def extra_repr(self) -> str:
        return "\n".join(f'{a}={getattr(self, a)}'
                         for a in ['num_groups', 'use_mod', 'group_scaling', 'save_last'])


class QKVBimodalCSRPool(nn.Module, _SaveLast):
    """Key-Query attention over views: queries from the main modality, keys from the mapping
    features (optionally mixed with modality features), values from the modality
    (reference pooling.py:322-551; named AttentiveBimodalCSRPool in stale configs)."""

    def __init__(
            self, in_main=None, in_map=None, in_mod=None, out_mod=None,
            num_groups=1, use_mod_q=False, use_mod_k=False, nc_qk=8,
            gating=True, dim_scaling=True, group_scaling=False, debug=False,
            save_last=False, nc_inner=32, map_encoder='DeepSetFeat', **kwargs):
        super().__init__()
        self.nc_inner = nc_inner
        self._init_save_last(save_last, extra=('Q', 'K', 'C', 'A', 'G'))
        self.debug = debug
        if debug:
            group_scaling, dim_scaling, nc_qk, in_map, in_mod = False, True, 1, 1, None

        assert 1 <= num_groups <= in_mod, \
            f"Number of groups must be between 1 and in_mod={in_mod}."
        out_mod = in_mod if out_mod is None else out_mod
        self.in_mod = in_mod
        self.out_mod = out_mod
        self.nc_qk = nc_qk
        self.use_mod_q = use_mod_q
        self.use_mod_k = use_mod_k
        self.num_groups = num_groups
        self.dim_scaling = dim_scaling
        self.group_scaling = group_scaling

        self.E_main = MLP([in_main, nc_inner, nc_inner], bias=False)
        self.E_map = getattr(_local_modules, map_encoder)(in_map, nc_inner, **kwargs)
        self.E_mod = MLP([in_mod, out_mod, out_mod], bias=False)
        if self.use_mod_q:
            in_mix, out_mix = nc_inner + out_mod, nc_inner
            mid_mix = nearest_power_of_2((in_mix + out_mix) / 2, out_mix * 2)
            self.E_mix_Q = MLP([in_mix, mid_mix, out_mix], bias=False)
        self.Q = nn.Linear(nc_inner, nc_qk * num_groups, bias=True)
        if self.use_mod_k:
            in_mix, out_mix = nc_inner + in_mod, nc_inner
            mid_mix = nearest_power_of_2((in_mix + out_mix) / 2, out_mix * 2)
            self.E_mix_K = MLP([in_mix, mid_mix, out_mix], bias=False)
        self.K = nn.Linear(nc_inner, nc_qk * num_groups, bias=True)
        self.G = Gating(num_groups, bias=True) if gating else None

    def forward(self, x_main, x_mod, x_map, csr_idx):
        if self.debug:
            x_mod = _materialize(x_mod)
            device = x_map.device
            x_map = torch.rand((x_map.shape[0], 1), device=device)
            idx_destroyed = torch.where(x_map < 0.3)[0]
            x_mod[idx_destroyed] = torch.rand(
                (idx_destroyed.shape[0], *(x_mod.shape[1:])), device=device)

        n_views = x_mod.shape[0]
        if fused_chain.keys_applicable(self, x_mod, x_map, csr_idx):
            # bf16 under autocast (round 4): the keys are one more layer of the recompute chain (one bf16 [V, 32] row per
            # view), compatibilities from them and the point's query row in one kernel; E_mod on the map rows, E_main
            # on the N point rows through the row kernels (split-K weight gradients: the library's are skinny GEMMs)
            x_main = _mlp_rows(self.E_main, x_main)
            x_mod = x_mod.with_rows(mlp_on_gathered_rows(self.E_mod, x_mod.rows, x_mod.counts, x_mod.shape[0]))
            # the rows the view kernel is handed are E_mod's OUTPUT: check those again (ADVICE r4 -- E_mod's torch
            # fallback may return fp32 rows, which the bf16 kernel would reinterpret), as GroupBimodalCSRPool does
            if fused_chain.keys_rows_ok(self, x_mod):
                if fused_chain.QKV_ONE_KERNEL and x_mod.rows.shape[1] in (32, 64, 128, 256, 512):
                    # keys, compatibilities, softmax, weighted sum and gate in ONE view kernel (dva_chain_attn_fwd_keys)
                    return fused_chain.qkv_pool(self, x_main, x_mod, x_map, csr_idx)
                compatibilities = fused_chain.qkv_compatibilities(self, x_main, x_map, csr_idx)
                x_pool, _, _ = _pool_with_attention(self, x_mod, compatibilities, csr_idx)
                return x_pool
            encoded = True                  # generic composition below, E_main / E_mod already applied
        else:
            encoded = False
        if not encoded:
            x_main = self.E_main(x_main)
        fused_keys = (not self.use_mod_k and not self.save_last and not self.debug
                      and fused_deepset.applicable(self.E_map, self.K, x_map))
        if fused_keys:
            keys = fused_deepset.deepset_linear(self.E_map, self.K, x_map, csr_idx)
        else:
            x_map = self.E_map(x_map, csr_idx)
        if encoded:
            pass
        elif isinstance(x_mod, ops.GatheredFeatures) and not (self.use_mod_k or self.use_mod_q or self.debug):
            x_mod = x_mod.with_rows(mlp_on_gathered_rows(self.E_mod, x_mod.rows, x_mod.counts, x_mod.shape[0]))
        else:
            x_mod = _mlp_rows(self.E_mod, _materialize(x_mod))

        if self.use_mod_k:
            keys = self.K(self.E_mix_K(torch.cat([x_map, x_mod.to(x_map.dtype)], dim=1)))
        elif not fused_keys:
            keys = self.K(x_map)

        if self.use_mod_q:
            # view-wise queries from the point features expanded to views
            x_main_q = gather_csr(x_main, csr_idx, n_rows=n_views)
            queries = self.Q(self.E_mix_Q(torch.cat([x_main_q, x_mod.to(x_main_q.dtype)], dim=1)))
        else:
            # point-wise queries expanded to views
            queries = gather_csr(self.Q(x_main), csr_idx, n_rows=n_views)

        compatibilities = (keys.reshape(n_views, self.num_groups, self.nc_qk)
                           * queries.reshape(n_views, self.num_groups, self.nc_qk)).sum(dim=2)
        if self.dim_scaling:
            compatibilities = compatibilities / math.sqrt(self.nc_qk)

        x_pool, attentions, gating = _pool_with_attention(self, x_mod, compatibilities, csr_idx)
        self._save(x_map, x_mod, csr_idx, K=keys, Q=queries, C=compatibilities, A=attentions,
                   **({'G': gating} if self.G is not None else {}))
        return x_pool

    def extra_repr(self) -> str:
        return "\n".join(f'{a}={getattr(self, a)}'
                         for a in ['dim_scaling', 'group_scaling', 'save_last'])


class MinMaxDiffSetFeat(nn.Module):
    """Element-wise set features from difference-to-min / difference-to-max / set size
    (reference pooling.py:554-601)."""

    def __init__(self, d_in, d_out, use_min=True, use_max=True, use_num=False, **kwargs):
        super().__init__()
        self.d_in = d_in
        self.d_out = d_out
        self.use_min = use_min
        self.use_max = use_max
        self.use_num = use_num
        in_mlp = d_in * (1 + self.use_min + self.use_max) + self.use_num
        self.mlp = MLP([in_mlp, d_out, d_out], bias=False)

    def forward(self, x, csr_idx):
        feats = [x]
        if self.use_min:
            feats.append(x - segment_gather_csr(x, csr_idx, reduce='min'))
        if self.use_max:
            feats.append(x - segment_gather_csr(x, csr_idx, reduce='max'))
        if self.use_num:
            sizes = csr_idx[1:] - csr_idx[:-1]
            num = torch.sqrt(1 / (sizes + 1e-3)).to(x.dtype)
            feats.append(gather_csr(num.view(-1, 1), csr_idx, n_rows=x.shape[0]))
        return self.mlp(torch.cat(feats, dim=1))

    def extra_repr(self) -> str:
        return "\n".join(f'{a}={getattr(self, a)}' for a in ['use_min', 'use_max', 'use_num'])


class DeepSetFeat(nn.Module):
    """DeepSets-style element features: elt MLP -> set pooling (+ set size) -> set MLP ->
    redistribute -> fuse -> elt MLP (reference pooling.py:604-673)."""

    _POOLING_MODES = ['max', 'mean', 'min', 'sum']
    _FUSION_MODES = ['residual', 'concatenation', 'both']

    def __init__(self, d_in, d_out, pool='max', fusion='concatenation', use_num=False, **kwargs):
        super().__init__()
        pool = pool.split('_')
        assert all([p in self._POOLING_MODES for p in pool]), \
            f"Unsupported pool='{pool}'. Expected elements of: {self._POOLING_MODES}"
        self.pool = pool
        if fusion not in self._FUSION_MODES:
            raise NotImplementedError(
                f"Unknown fusion='{fusion}'. Please choose among supported modes: "
                f"{self._FUSION_MODES}.")
        self.fusion = fusion
        self.d_in = d_in
        self.d_out = d_out
        self.use_num = use_num
        self.mlp_elt_1 = MLP([d_in, d_out, d_out], bias=False)
        self.mlp_set = MLP([d_out * len(self.pool) + self.use_num, d_out, d_out], bias=False)
        in_last_mlp = d_out if fusion == 'residual' else d_out * 2
        self.mlp_elt_2 = MLP([in_last_mlp, d_out, d_out], bias=False)

    def forward(self, x, csr_idx):
        n_rows = x.shape[0]
        x = self.mlp_elt_1(x)
        x_set = torch.cat([segment_csr(x, csr_idx, reduce=p) for p in self.pool], dim=-1)
        if self.use_num:
            # heuristic normalisation of the set size to [0, 1] (pooling.py:661-664)
            sizes = csr_idx[1:] - csr_idx[:-1]
            set_num = torch.sqrt(1 / (sizes + 1e-3)).to(x_set.dtype)
            x_set = torch.cat((x_set, set_num.view(-1, 1)), dim=1)
        x_set = gather_csr(self.mlp_set(x_set), csr_idx, n_rows=n_rows)
        if self.fusion == 'residual':
            x_out = x + x_set
        elif self.fusion == 'concatenation':
            x_out = torch.cat((x, x_set), dim=-1)
        else:
            x_out = torch.cat((x, x + x_set), dim=-1)
        return self.mlp_elt_2(x_out)

    def extra_repr(self) -> str:
        return "\n".join(f'{a}={getattr(self, a)}' for a in ['pool', 'fusion', 'use_num'])


class MLPSetFeat(nn.Module):
    """Element-wise features with a plain MLP (reference pooling.py:676-687)."""

    def __init__(self, d_in, d_out, **kwargs):
        super().__init__()
        self.d_in = d_in
        self.d_out = d_out
        self.mlp = MLP([d_in, d_out, d_out], bias=False)

    def forward(self, x, csr_idx):
        return self.mlp(x)


def nearest_power_of_2(x, min_power=16):
    """Nearest power of two of x, not below ``min_power`` (reference pooling.py:718-734;
    ties go to the larger power)."""
    x = int(x)
    if x < min_power:
        return min_power
    upper = 1 << (x - 1).bit_length()
    lower = upper >> 1
    return lower if x - lower < upper - x else upper


def group_sizes(num_elements, num_groups):
    """Sizes of ``num_groups`` near-equal groups partitioning ``num_elements`` channels
    (reference pooling.py:737-745): the first ``num_elements % num_groups`` groups get one more."""
    base, rem = divmod(num_elements, num_groups)
    sizes = torch.full((num_groups,), base, dtype=torch.long)
    sizes[:rem] += 1
    return sizes


def expand_group_feat(A, num_groups, num_channels):
    """Repeat each group column over the channels of its group (reference pooling.py:748-755)."""
    if num_groups == 1:
        A = A.view(-1, 1)
    elif num_groups < num_channels:
        A = A.repeat_interleave(group_sizes(num_channels, num_groups).to(A.device), dim=1)
    return A
