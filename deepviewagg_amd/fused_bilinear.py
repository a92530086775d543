"""GroupBimodalCSRPool on a lazily BILINEAR-gathered bf16 feature map (``interpolate=True``: the reference's
published KITTI-360 configuration, conf/models/segmentation/multimodal/sparseconv3d.yaml:7269-7340) through the
recompute chain with a per-view E_mod (``csrc/chain_emod.hip``, C ABI ``dva_emod_*``).

Reference dataflow: ``sparse_interpolation`` (core/multimodal/image.py:105-170) -> [V, C] -> atomic pool (identity for
an exact mapping) -> ``GroupBimodalCSRPool.forward`` (modules/multimodal/pooling.py:263-315): E_mod = MLP([in_mod,
out_mod, out_mod]) on the V rows, DeepSetFeat scores, softmax over the views of a point, weighted sum, gate.
Here: the first Linear of E_mod commutes with the interpolation and runs as ONE GEMM on the R map rows
(``Y = rows W_a^T``); BatchNorm_a, LeakyReLU, Linear_b (on the matrix cores), BatchNorm_b, LeakyReLU are evaluated per
view inside every pass that needs the values.  Eval mode is one kernel on the four taps of Y.  In train mode the first
statistics pass keeps the interpolated row ``z_a`` as bf16 [V, C_out] (the rounding Linear_a's output has under
autocast) and the later passes read it instead of gathering the taps again (the gathers bounded every pass).
Train-mode BatchNorm of E_mod adds two statistics passes; the backward hands ONE bf16 [V, C_out] gradient (dy_a -> dz_a
in place) to the weighted segmented reduction over the row plan of the taps (``dva_gather_rows_sum``: deterministic,
no atomics; views grouped by the anchor of their 2 x 2 tap block), which yields the gradient of Y; Linear_a's backward is autograd on the map rows.
The DeepSetFeat chain (scores) is shared with ``fused_chain`` (``chain_prologue`` / ``chain_epilogue``).
"""
import os

import torch

from . import _lib, ops, fused_chain, fused_deepset
from ._lib import check, ptr, require_device, stream_of
from .fused_chain_bwd import Arena, chain_epilogue

_POS = {}
# BatchNorm_a backward inside the anchor scatter: at the level of the anchor (Gram matrix of the tap weights x the four
# rows of Y: one random row per view) instead of row by row from the stored z_a (two)
ANCHOR_GRAM = os.environ.get("DVA_ANCHOR_GRAM", "1") == "1"
# the first statistics pass (taps of Y -> z_a) in anchor order (round 4): 0 = in view order over the tile table (round 3)
ANCHOR_ORDER_STATS = os.environ.get("DVA_ANCHOR_ORDER_STATS", "1") == "1"


def position_order(C, device):
    """kappa [C]: channel held by position p of the kernels' row layout (32 b + 16 h + r -> 32 b + chan(r, h))."""
    key = (C, str(device))
    if key not in _POS:
        p = torch.arange(C)
        b, h, r = p // 32, (p % 32) // 16, p % 16
        _POS[key] = (32 * b + (r & 3) + 8 * (r >> 2) + 4 * h).to(device)
    return _POS[key]


def _emod_blocks(module):
    e = module.E_mod
    if len(e) != 2:
        return None
    return e[0][0], e[0][1].batch_norm, e[0][2], e[1][0], e[1][1].batch_norm, e[1][2]


def applicable(module, x_mod, x_map, csr_idx):
    """Can ``module`` (a GroupBimodalCSRPool) pool ``x_mod`` (an ops.InterpolatedFeatures) on the fused path?"""
    if not fused_chain.enabled() or module.use_mod or module.save_last:
        return False
    if not isinstance(x_mod, ops.InterpolatedFeatures) or not x_mod.exact or x_mod.rows.dtype != torch.bfloat16:
        return False
    if not fused_deepset.applicable(module.E_map, module.E_score, x_map):
        return False
    if module.G is not None and any(t is not None and t.dtype != torch.float32 for t in (module.G.weight, module.G.bias)):
        return False
    blocks = _emod_blocks(module)
    if blocks is None:
        return False
    lin_a, bn_a, act_a, lin_b, bn_b, act_b = blocks
    C, G = module.out_mod, module.num_groups
    if C not in (32, 64, 128, 256) or G not in (1, 2, 4) or C % G or (C // G) % 8 or (C == 256 and G != 4):
        return False
    # (C_out = 128 / 256 -- the KITTI-360 pyramid levels 256 -> 128, 512 -> 256 -- train on the block-by-block kernels since
    #  round 4: chain_emod.hip "wide rows")
    if lin_a.bias is not None or lin_b.bias is not None or lin_b.in_features != C or lin_b.out_features != C:
        return False
    for bn, act in ((bn_a, act_a), (bn_b, act_b)):
        if not bn.affine or not bn.track_running_stats or bn.momentum is None \
                or getattr(act, 'negative_slope', None) != 0.2 or bn.training != module.E_map.training:
            return False
        if any(t.dtype != torch.float32 for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)):
            return False
    if lin_a.weight.dtype != torch.float32 or lin_b.weight.dtype != torch.float32:
        return False
    V, R = x_mod.shape[0], x_mod.rows.shape[0]
    N = csr_idx.shape[0] - 1
    return V == x_map.shape[0] and size_limits_ok(V, R, N, C)


def size_limits_ok(V, R, N, C):
    """The kernels address the view-sized arrays of a step with 32-bit byte offsets into one buffer descriptor: the tap
    table [V, 4] int32, the 64-byte rows the DeepSetFeat chain hands between its backward passes (x_map rows are 32
    bytes), the map rows Y [R, C] bf16 and the per-point rows [N, max(2 C, 128)].  z_a / dy_a [V, C] bf16 take one
    descriptor per TILE and have no such limit (4 GiB at V = 2^25, C = 64)."""
    lim = (1 << 32) - 16
    return 4 * V < (1 << 31) and V * 64 < lim and R * C * 2 < lim and N * max(C * 2, 128) < lim


class _EmodPool(torch.autograd.Function):
    """inputs: Y [R, C] (position order), taps, x_map, csr_idx; params: gamma_a, beta_a, W_b, gamma_b, beta_b, then the
    chain's parameters in fused_chain.chain_params order."""

    @staticmethod
    def forward(ctx, Y, rows4, w4, anchors, bhw, x_map, csr_idx, module, scaling, eps, ga_, ba_, Wb, gb_, bb_, *chain_p):
        lib = _lib.load()
        require_device(Y, rows4, w4, x_map, csr_idx)
        Y = Y.contiguous()
        x_map = x_map.contiguous()
        dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
        R, C = Y.shape
        st = stream_of(x_map)
        _, bn_a, _, _, bn_b, _ = _emod_blocks(module)
        S = fused_chain.chain_prologue(module, x_map, csr_idx)
        training = S.training
        G = S.G
        eops = torch.empty(2 * (C // 32) ** 2 * 2 * 1024, dtype=torch.uint8, device=dev)
        check(lib.dva_emod_prep(ptr(Wb.detach().float().contiguous()), C, ptr(eops), st), "dva_emod_prep")
        tap_bytes = V * (4 * C * 2 + 32)
        # train mode: z_a (the interpolated Linear_a output, rounded to bf16 like the layer's output under autocast) is
        # written once by the first statistics pass; every later pass of the step reads these 2 C bytes per view instead
        # of gathering 4 taps x 2 C bytes again.  Eval mode is one kernel and evaluates the taps itself.
        za = torch.empty((V, C), dtype=torch.bfloat16, device=dev) if training else None
        za_bytes = V * C * 2

        # train mode: the first statistics pass walks the views in ANCHOR order (the plan the backward scatters through,
        # built here once): neighbouring lanes then read the same four rows of Y and the tap gathers become cache hits
        # (C_out = 32: one block per view, nothing to overlap the random record reads with -- 2.4 against 1.9 ms in view order)
        plan = ops.anchor_plan(anchors, bhw) if (training and ANCHOR_ORDER_STATS and C >= 64) else None

        def stats(layer, tab_a):
            s = ops.zeros_small(2 * C, torch.float64, dev)
            if training and layer == 1 and plan is not None:
                with ops._timed("emod_stats1", V * (4 + 32 + C * 2)):
                    check(lib.dva_emod_stats1_plan(ptr(Y), ptr(rows4), ptr(w4), ptr(plan[0]), ptr(s), ptr(za), V, R, C,
                                                   st), "dva_emod_stats1_plan")
            elif training:
                with ops._timed(f"emod_stats{layer}", tap_bytes + za_bytes if layer == 1 else za_bytes):
                    check(lib.dva_emod_stats(layer, ptr(Y), ptr(rows4), ptr(w4), ptr(S.tiles), ptr(S.n_tiles), ptr(eops),
                                             ptr(tab_a), ptr(s), ptr(za), V, R, C, st), "dva_emod_stats")
            return s
        tab_a = ops.bn_table(stats(1, None), float(max(V, 1)), bn_a, training)
        tab_b = ops.bn_table(stats(2, tab_a), float(max(V, 1)), bn_b, training)
        out = fused_chain.pooled_output(csr_idx, N, C, dev)
        need_bwd = any(ctx.needs_input_grad)
        scores = torch.empty((V, 4), dtype=torch.float32, device=dev) if need_bwd else None
        # per view: z_a (train) or 4 taps of Y (C s each) + tap record 32 (eval), x_map 32, view -> point index 4
        # (+ scores 16 out in training)
        with ops._timed("emod_attn_fwd", (za_bytes if training else tap_bytes)
                        + V * (32 + 4 + (16 if need_bwd else 0)) + N * (C * 2 + 128 + 8)):
            check(lib.dva_emod_attn_fwd(ptr(x_map), ptr(S.vp), ptr(S.t_add), ptr(S.tiles), ptr(S.n_tiles), ptr(S.wops),
                                        ptr(S.bn1), ptr(S.bn2), ptr(S.bn5), ptr(S.bn6), ptr(S.bs), ptr(Y), ptr(rows4),
                                        ptr(w4), ptr(eops), ptr(tab_a), ptr(tab_b), ptr(csr_idx), ptr(S.gw), ptr(S.gb),
                                        ptr(out), ptr(scores), ptr(za), N, V, R, C, G, int(scaling), float(eps), st),
                  "dva_emod_attn_fwd")
        ctx.save_for_backward(Y, rows4, w4, x_map, csr_idx, S.vp, S.tiles, S.n_tiles, S.wops, S.t_add, S.zstar, S.arg,
                              S.mom, S.bn1, S.bn2, S.bn5, S.bn6, out, scores, S.bs, S.gw, S.gb, S.W1, eops, tab_a, tab_b,
                              za)
        ctx.module = module
        ctx.set_saved = S.set_saved
        ctx.training = training
        ctx.meta = (int(scaling), float(eps))
        ctx.anchors, ctx.bhw = anchors, bhw
        ctx.anchor_plan = plan
        return out

    @staticmethod
    def backward(ctx, gout):
        from types import SimpleNamespace
        lib = _lib.load()
        if ctx.set_saved is None:
            raise RuntimeError("the recompute chain's backward ran twice on the same graph (retain_graph is not "
                               "supported on this path)")
        (Y, rows4, w4, x_map, csr_idx, vp, tiles, n_tiles, wops, t_add, zstar, arg, mom, bn1, bn2, bn5, bn6, out,
         scores, bs, gw, gb, W1, eops, tab_a, tab_b, za) = ctx.saved_tensors
        module, training = ctx.module, ctx.training
        scaling, eps = ctx.meta
        gate = module.G
        dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
        R, C = Y.shape
        G = module.E_score.weight.shape[0]
        st = stream_of(x_map)
        gout = gout.contiguous().to(torch.bfloat16)
        inv_m = (1.0 / float(max(V, 1))) if training else 0.0
        S = SimpleNamespace(vp=vp, tiles=tiles, n_tiles=n_tiles, wops=wops, t_add=t_add, zstar=zstar, arg=arg, mom=mom,
                            bn1=bn1, bn2=bn2, bn5=bn5, bn6=bn6, W1=W1, G=G, training=training)
        arena = Arena(dev, n_floats=(1 << 15) + C * C)      # + dW_b (65 536 floats at C_out = 256)
        za_bytes = V * C * 2
        if za is None:      # forward ran in eval mode: the backward passes read the stored z_a, build it now
            za = torch.empty((V, C), dtype=torch.bfloat16, device=dev)
            scratch = torch.zeros(2 * C, dtype=torch.float64, device=dev)
            check(lib.dva_emod_stats(1, ptr(Y), ptr(rows4), ptr(w4), ptr(tiles), ptr(n_tiles), None, None, ptr(scratch),
                                     ptr(za), V, R, C, st), "dva_emod_stats")

        def consts(stats, tab):
            sm, dg, db = arena.take(2 * C), arena.take(C), arena.take(C)
            check(lib.dva_bn_bwd_consts(ptr(stats), ptr(tab), inv_m, 1, ptr(sm), ptr(dg), ptr(db), C, st),
                  "dva_bn_bwd_consts")
            return sm, dg, db
        # ---- attention + gate backward (scores from the forward, E_mod re-evaluated): dc, view records, S of BatchNorm_b
        dc = torch.empty((V, 4), dtype=torch.float32, device=dev)
        rec = torch.empty((V, 4), dtype=torch.int32, device=dev)
        gwb = arena.take(2 * G) if gate is not None else None
        stats_b = ops.zeros_small(2 * C, torch.float64, dev)
        with ops._timed("emod_attn_bwd", za_bytes + V * (16 + 4 + 16 + 16) + N * (C * 2 + 8)):
            check(lib.dva_emod_attn_bwd(ptr(scores), ptr(vp), ptr(tiles), ptr(n_tiles), None, None, None,
                                        ptr(eops), ptr(tab_a), ptr(tab_b), ptr(csr_idx), ptr(gw), ptr(gb), ptr(gout),
                                        ptr(out), ptr(dc), ptr(rec), ptr(gwb), ptr(stats_b), ptr(za), N, V, R, C, G,
                                        scaling, eps, st), "dva_emod_attn_bwd")
        del scores
        sm_b, dg_b, db_b = consts(stats_b, tab_b)
        # ---- E_mod backward: dW_b, dy_a handed over as bf16 [V, C], S of BatchNorm_a; then dz_a in place
        da = torch.empty((V, C), dtype=torch.bfloat16, device=dev)
        dWb = arena.take(C, C)
        stats_a = ops.zeros_small(2 * C, torch.float64, dev)
        def emod_bwd(stage, name, nbytes):
            with ops._timed(name, nbytes):
                check(lib.dva_emod_bwd(stage, None, None, None, ptr(tiles), ptr(n_tiles), ptr(eops), ptr(tab_a),
                                       ptr(tab_b), None, ptr(sm_b), ptr(rec), ptr(gout), ptr(da), ptr(dWb),
                                       ptr(stats_a), ptr(za), N, V, R, C, G, st), "dva_emod_bwd")
        if C == 128:      # wide rows: dy_a + S of BatchNorm_a, then dW_b (Linear_b evaluated once more), two kernels
            emod_bwd(3, "emod_bwd_dya", za_bytes + V * (16 + C * 2) + N * C * 2)
            emod_bwd(4, "emod_bwd_wgrad", za_bytes + V * 16 + N * C * 2)
        elif C == 256:    # dz_b -> da, dW_b from the stored dz_b, dy_a in place: three kernels behind stage 2
            emod_bwd(2, "emod_bwd_b", 3 * za_bytes + V * (16 + 4 * C * 2) + N * C * 2)
        else:
            emod_bwd(2, "emod_bwd_b", za_bytes + V * (16 + C * 2) + N * C * 2)
        del rec
        sm_a, dg_a, db_a = consts(stats_a, tab_a)
        # ---- gradient of Y: the transpose of the interpolation applied to dz_a = BatchNorm_a backward of dy_a (views
        #      grouped by the anchor of their 2 x 2 tap block: every row read once, deterministic; the BatchNorm
        #      backward is applied to the rows as they are read: no in-place pass over [V, C])
        dY = None
        if ctx.needs_input_grad[0]:
            dY = ops.bilinear_scatter(da, rows4, w4, ctx.anchors, ctx.bhw, plan=ctx.anchor_plan,
                                      bn_backward=(za, tab_a, sm_a, Y if ANCHOR_GRAM else None)).to(Y.dtype)
        del da, za
        grads = chain_epilogue(lib, arena, S, module, x_map, csr_idx, dc, gwb, ctx.set_saved)
        ctx.set_saved = None
        return (dY, None, None, None, None, None, None, None, None, None, dg_a, db_a, dWb, dg_b, db_b) + tuple(grads)


def pool(module, x_mod, x_map, csr_idx):
    """``module`` = GroupBimodalCSRPool, ``x_mod`` = ops.InterpolatedFeatures of the raw feature maps."""
    lin_a, bn_a, _, lin_b, bn_b, _ = _emod_blocks(module)
    C = module.out_mod
    kappa = position_order(C, lin_a.weight.device)
    # Linear_a on the map rows, its output channels in the kernels' position order (autograd: index + GEMM)
    Y = ops.tall_linear(x_mod.rows, lin_a.weight[kappa])
    csr_idx = ops._check_ptr(csr_idx)
    # (one (B, H, W) per setting: a multi-setting batch -- ops.InterpolatedFeatures.cat -- is one gather over stacked rows)
    return _EmodPool.apply(Y, x_mod.tap_rows, x_mod.tap_weights, x_mod.anchors, tuple(x_mod.geometry), x_map, csr_idx,
                           module,
                           module.group_scaling, 1e-12,
                           bn_a.weight, bn_a.bias, lin_b.weight, bn_b.weight, bn_b.bias,
                           *fused_chain.chain_params(module))
