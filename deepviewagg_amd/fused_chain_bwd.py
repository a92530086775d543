"""Backward of fused_chain._ChainPool: four view passes (attention, layer 6, layer 5, layer 2), each re-evaluating
the DeepSetFeat chain from x_map, separated by the BatchNorm-backward statistics; the per-point set branch in
between runs on csrc/chain_set.hip (fused_chain._set_branch_backward)."""
import os

import torch

from . import _lib, ops
from ._lib import check, ptr, stream_of
from .fused_deepset import D


class Arena:
    """One zero-filled fp32 buffer handed out in 16-byte aligned pieces: the small accumulators and gradients of one
    backward cost one fill launch instead of one each."""

    def __init__(self, device, n_floats=1 << 15):
        self.buf = ops.zeros_small(n_floats, torch.float32, device)
        self.used = 0

    def take(self, *shape):
        n = 1
        for d in shape:
            n *= d
        off = self.used
        self.used = off + (n + 3) // 4 * 4
        assert self.used <= self.buf.numel(), "Arena too small"
        return self.buf[off:off + n].view(*shape)


def bn_bwd_consts(lib, arena, stats, bn, m_rows, training, st, hat=True, out=True):
    sm = dg = db = None
    if out:
        sm, dg, db = arena.take(2 * D), arena.take(D), arena.take(D)
    check(lib.dva_bn_bwd_consts(ptr(stats), ptr(bn), (1.0 / m_rows) if training else 0.0, 1 if hat else 0,
                                ptr(sm), ptr(dg), ptr(db), D, st), "dva_bn_bwd_consts")
    return sm, dg, db


# DVA_CHAIN_MERGE=1 -- merged backward (round 5): the score pass also sums what the statistics of the
# BatchNorm-5 backward are linear in, stage 6 disappears and stage 5 starts from the score gradients (csrc/chain_bwd.hip
# score_l6_kernel).  Built, parity-green (tests/test_gpu_chain.py::test_merged_backward_matches_three_pass and the whole
# chain / bilinear / pooling / full-size suites under the switch), put on a register diet until both merged passes ran at
# three wavefronts per SIMD, and measured: the step does not get faster (11.01 against 10.92 - 10.95 ms on one box:
# profiles/r05_chain_merge_ab.json) -- the per-tile bookkeeping of the merge costs as many vector instructions as the chain
# evaluation of stage 6 it removes.  Off by default.
MERGE_STAGE6 = os.environ.get("DVA_CHAIN_MERGE", "0") == "1"


def chain_epilogue(lib, arena, S, module, x_map, csr_idx, dc, gwb, set_saved, keys=None):
    """Everything of a chain backward behind the attention backward (which is specific to how the values are
    produced): score layer + BatchNorm-6 statistics, the three layer passes, the per-point set branch, layer 1.
    ``S``: namespace with vp, tiles, n_tiles, wops, t_add, zstar, arg, mom, bn1, bn2, bn5, bn6, W1, G, training.
    ``dc`` fp32 [V, 4] score gradients (consumed), ``gwb`` fp32 [2 G] gate gradients or None.
    ``keys`` = (Qp fp32 [N, 32], groups, scale): the last layer is the KEY layer of a QKVBimodalCSRPool (S.G = 32) and
    ``dc`` the gradient of the compatibilities [V, 4]; the passes build dK' = scale dc[g] Q'[point] in registers.
    Returns the gradients in the order of fused_chain.chain_params(module)."""
    from .fused_chain import _set_branch_backward
    e_map, gate = module.E_map, module.G
    dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
    G, training = S.G, S.training
    vp, tiles, n_tiles, wops, t_add = S.vp, S.tiles, S.n_tiles, S.wops, S.t_add
    bn1, bn2, bn5, bn6 = S.bn1, S.bn2, S.bn5, S.bn6
    st = stream_of(x_map)
    m_rows = float(max(V, 1))
    zpool = iter(ops.zeros_small((10, 2 * D), torch.float64, dev))

    def zstats():
        return next(zpool)

    def consts(stats, bn, hat=True, out=True):
        """The arithmetic between two passes in one launch (dva_bn_bwd_consts): S2 -> z_hat form in place,
        then (sm = S / M for the next pass, d gamma = S2, d beta = S1)."""
        return bn_bwd_consts(lib, arena, stats, bn, m_rows, training, st, hat, out)

    # ---- score layer: dWs, dbs, and the statistics of the BatchNorm-6 backward (one chain evaluation)
    s6 = zstats()
    dWs, dbs = arena.take(G, D), arena.take(G)
    merged = MERGE_STAGE6 and keys is None
    a2 = getattr(S, "a2", None) if (keys is None and not merged) else None      # the stored-a2 hybrid (fused_chain.CHAIN_A2)
    xb = 64 if a2 is not None else 32          # bytes per view of the pass's chain input
    with ops._timed("chain_score_stats", V * (xb + 4 + 16) + N * (128 if keys is None else 256)):
        if merged:
            # + the sums the statistics of the BatchNorm-5 backward are linear in: P2 | Q2 [2, 32, 32], e1 | e2 | n5 | q5
            acc5 = arena.take(2, D, D)
            vec5 = ops.zeros_small(4 * D, torch.float64, dev)
            check(lib.dva_chain_score_l6_stats(ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                               ptr(bn1), ptr(bn2), ptr(bn5), ptr(bn6), ptr(dc), ptr(s6), ptr(dWs),
                                               ptr(dbs), ptr(acc5), ptr(vec5), G, V, N, st), "dva_chain_score_l6_stats")
        elif keys is None and a2 is not None:
            check(lib.dva_chain_score_stats_a2(ptr(a2), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops), ptr(bn5),
                                               ptr(bn6), ptr(dc), ptr(s6), ptr(dWs), ptr(dbs), G, V, N, st),
                  "dva_chain_score_stats_a2")
        elif keys is None:
            check(lib.dva_chain_score_stats(ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                            ptr(bn1), ptr(bn2), ptr(bn5), ptr(bn6), ptr(dc), ptr(s6), ptr(dWs), ptr(dbs),
                                            G, V, N, st), "dva_chain_score_stats")
        else:
            Qp, qk_groups, qk_scale = keys
            check(lib.dva_chain_score_stats_keys(ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                                 ptr(bn1), ptr(bn2), ptr(bn5), ptr(bn6), ptr(dc), ptr(Qp), ptr(s6),
                                                 ptr(dWs), ptr(dbs), qk_groups, qk_scale, V, N, st),
                  "dva_chain_score_stats_keys")

    def layer(stage, sm2, sm5, sm6, arg_, dpooled_, da_in, da_out, dW, du, P, stats, name, nbytes):
        with ops._timed(name, nbytes):
            if keys is not None and stage == 6:
                check(lib.dva_chain_bwd_layer6_keys(ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                                    ptr(bn1), ptr(bn2), ptr(bn5), ptr(bn6), ptr(sm6), ptr(dc), ptr(keys[0]),
                                                    ptr(da_out), ptr(dW), ptr(stats), keys[1], keys[2], V, N, st),
                      "dva_chain_bwd_layer6_keys")
                return
            if stage == 6 and a2 is not None:
                check(lib.dva_chain_bwd_layer6_a2(ptr(a2), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                                  ptr(bn5), ptr(bn6), ptr(sm6), ptr(dc), ptr(da_out), ptr(dW), ptr(stats),
                                                  min(G, 4), V, N, st), "dva_chain_bwd_layer6_a2")
                return
            check(lib.dva_chain_bwd_layer(stage, ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                          ptr(bn1), ptr(bn2), ptr(bn5), ptr(bn6), ptr(sm2), ptr(sm5), ptr(sm6),
                                          ptr(dc), ptr(arg_), ptr(dpooled_), ptr(da_in), ptr(da_out), ptr(dW),
                                          ptr(du), ptr(P), ptr(stats), min(G, 4), V, N, st),
                  "dva_chain_bwd_layer")

    # per view: x_map 32 + view->point 4 (+ score gradients 16) + the 64-byte gradient row handed between the passes
    sm6, g6, b6 = consts(s6, bn6)
    dW6 = arena.take(D, D)
    s5 = zstats()
    if merged:
        W6 = module.E_map.mlp_elt_2[1][0].weight.detach().float().contiguous()
        check(lib.dva_chain_l6_consts(ptr(sm6), ptr(bn6), ptr(W6), ptr(acc5), ptr(vec5), ptr(s5), st),
              "dva_chain_l6_consts")
        da5 = None
    else:
        da5 = torch.empty((V, D), dtype=torch.bfloat16, device=dev)
        layer(6, None, None, sm6, None, None, None, da5, dW6, None, None, s5, "chain_bwd_l6",
              V * (xb + 4 + 16 + 64) + N * 128)
        del a2
    sm5, g5, b5 = consts(s5, bn5)
    dW5 = arena.take(D, 2 * D)
    du = torch.zeros((N, D), dtype=torch.float32, device=dev)
    s2 = zstats()
    da2 = torch.empty((V, D), dtype=torch.bfloat16, device=dev)
    if merged:
        # per view: x_map 32 + view -> point 4 + score gradients 16 in, the 64-byte dy2 row out
        with ops._timed("chain_bwd_l5", V * (32 + 4 + 16 + 64) + N * 256):
            check(lib.dva_chain_bwd_layer5_merged(ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                                  ptr(bn1), ptr(bn2), ptr(bn5), ptr(bn6), ptr(sm5), ptr(sm6), ptr(dc),
                                                  ptr(da2), ptr(dW5), ptr(dW6), ptr(du), ptr(s2), min(G, 4), V, N, st),
                  "dva_chain_bwd_layer5_merged")
    else:
        layer(5, None, sm5, None, None, None, da5, da2, dW5, du, None, s2, "chain_bwd_l5",
              V * (32 + 4 + 64 + 64) + N * 256)
    del da5
    # ---- per-point set branch
    dpooled, d_set = _set_branch_backward(set_saved, du, dW5, training, zstats, arena)
    consts(s2, bn2, out=False)            # view part; the per-point part below is accumulated in z_hat directly
    dpooled_dy = torch.empty((N, D), dtype=torch.float32, device=dev)     # leaky'(y*) dpooled: what stage 2 routes
    check(lib.dva_chain_route_stats(ptr(S.zstar), ptr(dpooled), ptr(bn2), ptr(csr_idx), ptr(s2), ptr(dpooled_dy), N,
                                    st), "dva_chain_route_stats")
    sm2, g2, b2 = consts(s2, bn2, hat=False)
    dW2, P = arena.take(D, D), arena.take(D, 20)       # P = sum dy1 [x_hi | x_lo | 1]^T
    s1 = zstats()
    layer(2, sm2, None, None, S.arg, dpooled_dy, da2, None, dW2, None, P, None, "chain_bwd_l2",
          V * (32 + 4 + 64) + N * 256)
    del da2
    check(lib.dva_chain_stats1(ptr(P), ptr(S.W1), 0, ptr(s1), st), "dva_chain_stats1")    # layer 1 is linear in x_map
    sm1, g1, b1 = consts(s1, bn1)
    # ---- first layer: BatchNorm-1 backward is linear in its statistics and z1 = W1 x is linear in x, so
    #      dW1 = G1 (P - (S1/M) SX^T - (S2/M) . Q) with Q = sum_v z1_hat x^T from the moments of x_map
    dW1 = arena.take(D, 8)
    check(lib.dva_chain_dw1(ptr(P), ptr(S.mom), ptr(S.W1), 0, ptr(bn1), ptr(sm1), ptr(dW1), st), "dva_chain_dw1")
    if gate is not None:
        gG = G if keys is None else keys[1]        # gate entries = attention groups (the key layer has 32 outputs)
        dgw, dgb = gwb[:gG].reshape(gate.weight.shape), gwb[gG:].reshape(gate.bias.shape)
    else:
        dgw = dgb = None
    return [dW1, g1, b1, dW2, g2, b2, dW5, g5, b5, dW6, g6, b6, dWs, dbs, dgw, dgb] + d_set


# DVA_OVERLAP_ROWS_GRAD=1: the rows gradient (random line fetches, few vector instructions) on a side stream,
# concurrently with the first passes of the chain epilogue (vector-unit bound) on the caller's stream: both only need the
# attention backward's outputs.  Measured on S1 (same box, A/B): 11.60 -> 11.43 ms/step -- the two share the CUs, each
# stretches (rows gradient 1.32 -> 2.12 ms, stage 6 1.22 -> 2.18 ms), the sum shrinks by 0.16 ms.  Off by default: the
# per-kernel durations of the bench line (and its roofline object) are only meaningful when kernels do not overlap.
OVERLAP_ROWS_GRAD = os.environ.get('DVA_OVERLAP_ROWS_GRAD', '0') == '1'
# A/B of round 4: the attention backward writes its 16-byte view records in PLAN order (slot =
# position of the view in the row plan, through dva_plan_inverse) so that the rows gradient streams them.  Measured
# (profiles/r04*_rows_grad_planrec_ab.json); off by default.
PLAN_ORDER_RECORDS = os.environ.get('DVA_ROWS_GRAD_PLANREC', '0') == '1'
_SIDE = {}


def _side_stream(dev):
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream(device=dev)
    return _SIDE[dev]


def backward(ctx, gout):
    from types import SimpleNamespace
    lib = _lib.load()
    if ctx.set_saved is None:
        raise RuntimeError("the recompute chain's backward ran twice on the same graph: its per-step workspaces are "
                           "released after the first backward (retain_graph is not supported on this path)")
    qk = getattr(ctx, "qk", None)          # QKVBimodalCSRPool in the view kernel: (groups, scale); two more saved tensors
    saved = ctx.saved_tensors
    (rows, row_idx, x_map, csr_idx, vp, tiles, n_tiles, wops, t_add, zstar, arg, mom,
     bn1, bn2, bn5, bn6, out, scores, bs, gw, gb, W1) = saved[:22]
    module, training = ctx.module, ctx.training
    scaling, eps = ctx.meta
    gate = module.G
    dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
    R, C = rows.shape
    G = qk[0] if qk is not None else module.E_score.weight.shape[0]
    st = stream_of(x_map)
    gout = gout.contiguous().to(torch.bfloat16)
    S = SimpleNamespace(vp=vp, tiles=tiles, n_tiles=n_tiles, wops=wops, t_add=t_add, zstar=zstar, arg=arg, mom=mom,
                        bn1=bn1, bn2=bn2, bn5=bn5, bn6=bn6, W1=W1, G=G if qk is None else D, training=training,
                        a2=getattr(ctx, "a2", None))
    ctx.a2 = None
    arena = Arena(dev)          # every small fp32 accumulator / gradient of this backward: one zero fill
    # ---- attention + gate backward from the scores the forward left: score gradients, view records (no chain)
    dc = torch.empty((V, 4), dtype=torch.float32, device=dev)
    rec = torch.empty((V, 4), dtype=torch.int32, device=dev)       # 16-byte records: point | 4 x bf16 weight | pad
    gwb = arena.take(2 * G) if gate is not None else None
    # per view: value row + scores 16 + view->point / row index 8 in, score gradients 16 + record 16 out; per point
    # grad_out row (+ out row for points with more than 32 views)
    plan = None
    if ctx.needs_input_grad[0]:
        plan = ctx.plan if ctx.plan is not None else ops.row_plan(row_idx, R, with_counts=False)[0]
    planrec = PLAN_ORDER_RECORDS and plan is not None and not isinstance(plan, ops.SplitPlan)
    if planrec:
        # A/B (round 4): records written in plan order through the inverse of the plan permutation
        inv = torch.empty(V, dtype=torch.int32, device=dev)
        with ops._timed("plan_inverse", V * 8):
            check(lib.dva_plan_inverse(ptr(plan[0]), ptr(inv), V, st), "dva_plan_inverse")
    with ops._timed("chain_attn_bwd", V * (C * 2 + 16 + 8 + 16 + 16) + N * (C * 2 + 8)):
        if planrec:
            check(lib.dva_chain_attn_bwd_planrec(ptr(inv), ptr(scores), ptr(vp), ptr(tiles), ptr(n_tiles), ptr(rows),
                                                 ptr(row_idx), ptr(csr_idx), ptr(gw), ptr(gb), ptr(gout), ptr(out),
                                                 ptr(dc), ptr(rec), ptr(gwb), N, V, R, C, G, scaling, eps, st),
                  "dva_chain_attn_bwd_planrec")
        else:
            check(lib.dva_chain_attn_bwd(ptr(scores), ptr(vp), ptr(tiles), ptr(n_tiles), ptr(rows), ptr(row_idx),
                                         ptr(csr_idx), ptr(gw), ptr(gb), ptr(gout), ptr(out), ptr(dc), ptr(rec),
                                         ptr(gwb), N, V, R, C, G, scaling, eps, st), "dva_chain_attn_bwd")
    del scores
    # ---- rows gradient: segmented reduction over the row plan (deterministic, no atomics)
    grows = None
    side = None
    if ctx.needs_input_grad[0]:
        split = isinstance(plan, ops.SplitPlan)
        perm, row_ptr = (None, plan.row_ptr) if split else plan

        def rows_grad(stream):
            # the row is rounded to the map's dtype where it is summed (round 5): no fp32 [R, C] tensor + conversion pass
            if split:     # the records themselves go through the plan's two scatter passes, then stream in plan order
                return ops.rows_grad_rec16(gout, plan, rec, R, C, G, rows.dtype, stream)
            g = torch.empty((R, C), dtype=rows.dtype, device=dev)
            with ops._timed("view_gather_rows_grad", V * (4 + 16 + C * 2) + R * (C * 2 + 4)):
                check(lib.dva_view_gather_rows_grad_rec16_to(ptr(gout), None if planrec else ptr(perm), ptr(row_ptr),
                                                             ptr(rec), ptr(g), _lib.dtype_code(g), R, V, C, G,
                                                             _lib.DVA_BF16, stream),
                      "dva_view_gather_rows_grad_rec16_to")
            return g
        if OVERLAP_ROWS_GRAD:
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            side.wait_stream(main)                       # records of the attention backward
            with torch.cuda.stream(side):
                grows = rows_grad(stream_of(x_map))
            for t_ in (gout, row_ptr, rec) + (() if perm is None else (perm,)):
                t_.record_stream(side)
        else:
            grows = rows_grad(st)
    del rec
    lead = (grows, None, None, None, None, None, None, None)
    keys_arg = None
    if qk is not None:
        # dQ' from the stored key rows; the key gradient itself is built inside the chain passes (chain_epilogue keys=...)
        keys_rows, Qp = saved[22], saved[23]
        dQ = torch.empty((N, D), dtype=torch.float32, device=dev)
        with ops._timed("qkv_dquery", V * (64 + 16) + N * 136):
            check(lib.dva_qkv_dquery(ptr(dc), 4, ptr(keys_rows), ptr(csr_idx), ptr(dQ), N, V, G, qk[1], st),
                  "dva_qkv_dquery")
        keys_arg = (Qp, G, qk[1])
        lead = lead + (dQ, None)
    grads = chain_epilogue(lib, arena, S, module, x_map, csr_idx, dc, gwb, ctx.set_saved, keys=keys_arg)
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
        grows.record_stream(torch.cuda.current_stream(dev))
    ctx.set_saved = None
    return lead + tuple(grads)
