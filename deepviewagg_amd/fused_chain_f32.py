"""DeepSetFeat + score layer for fp32 features outside autocast on the fp32 recompute chain
(``csrc/chain_f32.hip``, C ABI ``dva_chain3_*``): the drop-in of ``fused_deepset.deepset_linear`` for at most four
scores per view.

Computes ``linear(E_map(x_map, csr_idx))`` of the reference (modules/multimodal/pooling.py:658-669 DeepSetFeat.forward
followed by E_score :282) with every product on the fp32 matrix cores (exact fp32 fma chains), BatchNorm + LeakyReLU in
fp32 and fp64 statistics.  Same pass structure as the bf16 chain of ``fused_chain.py`` (one statistics pass per
BatchNorm layer forwards, one pass per BatchNorm-backward barrier backwards, the per-point set branch in between), but
balanced for fp32: these passes are bound by the matrix pipe (64 cycles per instruction), not by HBM, so the raw outputs
of layers 2 and 5 (fp32 [V, 32] each) stay in HBM and each pass starts from them instead of re-evaluating the chain
from ``x_map`` -- two stored tensors instead of the thirteen of the stored-activation kernels.
"""
from types import SimpleNamespace

import torch

from . import _lib, ops, fused_deepset
from ._lib import check, ptr, require_device, stream_of
from .fused_deepset import D, _bn_of
from .fused_chain import build_tiles, _chain_bn, _set_branch_forward, _set_branch_backward, chain_params
from .fused_chain_bwd import Arena, bn_bwd_consts

# False: the stored-activation passes of fused_deepset (tests, A/B)
ENABLED = True
OPS_BYTES = 26 * 1024          # 26 blocks of 64 float4 (dva_chain3_prep)
ROW = 128                      # bytes of one fp32 [., 32] row


def applicable(e_map, linear, x_map, csr_idx):
    """Can ``linear(e_map(x_map, csr_idx))`` run on the fp32 chain?  (fused_deepset.applicable + at most 4 scores per
    view + 32-bit buffer addressing of x_map and the per-point rows.)"""
    if not ENABLED or not fused_deepset.applicable(e_map, linear, x_map) or linear.out_features > 4:
        return False
    V, N = x_map.shape[0], csr_idx.shape[0] - 1
    return 0 < V and V * 32 < (1 << 32) - 16 and N * 128 < (1 << 32) - 16


class _ChainScores(torch.autograd.Function):
    """params in the order of fused_chain.chain_params (gate = None)."""

    @staticmethod
    def forward(ctx, x_map, csr_idx, shim, *params):
        lib = _lib.load()
        require_device(x_map, csr_idx)
        x_map = x_map.contiguous()
        e_map, e_score = shim.E_map, shim.E_score
        dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
        st = stream_of(x_map)
        training = e_map.training
        W1 = e_map.mlp_elt_1[0][0].weight.detach().contiguous()
        W2 = e_map.mlp_elt_1[1][0].weight.detach().contiguous()
        W5 = e_map.mlp_elt_2[0][0].weight.detach().contiguous()       # [32, 64]: per-view half | per-point half
        W6 = e_map.mlp_elt_2[1][0].weight.detach().contiguous()
        Ws, bs = e_score.weight.detach().contiguous(), e_score.bias.detach().contiguous()
        G = Ws.shape[0]
        bns = [_bn_of(e_map.mlp_elt_1[0]), _bn_of(e_map.mlp_elt_1[1]),
               _bn_of(e_map.mlp_elt_2[0]), _bn_of(e_map.mlp_elt_2[1])]
        zpool = iter(ops.zeros_small((10, 3 * D), torch.float64, dev))

        def zstats():
            return next(zpool)

        with ops._timed("chain_tiles", N * 8):
            tiles, n_tiles = build_tiles(csr_idx, V)
            vp = torch.empty(V, dtype=torch.int32, device=dev)
            check(lib.dva_csr_expand(ptr(csr_idx), N, ptr(vp), st), "dva_csr_expand")
        wops = torch.empty(OPS_BYTES, dtype=torch.uint8, device=dev)
        check(lib.dva_chain3_prep(ptr(W1), ptr(W2), ptr(W5), W5.shape[1], ptr(W6), ptr(Ws), G, ptr(wops), st),
              "dva_chain3_prep")
        # ---- layer 1: statistics from the moments of x_map (z1 = W1 x is linear in x)
        s1 = zstats()
        mom = ops.zeros_small(44, torch.float64, dev)
        if training:
            with ops._timed("chain_moments", V * 32):
                check(lib.dva_chain_moments(ptr(x_map), V, ptr(W1), 1, ptr(mom), ptr(s1), st), "dva_chain_moments")
        bn1 = _chain_bn(s1, V, bns[0], training)
        # ---- layer 2: statistics + set pooling; z2 stays
        s2 = zstats()
        zstar = torch.empty((N, D), dtype=torch.float32, device=dev)
        arg = torch.empty((N, D), dtype=torch.int32, device=dev)
        z2 = torch.empty((V, D), dtype=torch.float32, device=dev)
        with ops._timed("chain3_stats2", V * (36 + ROW) + N * 256):
            check(lib.dva_chain3_stats2(ptr(x_map), ptr(vp), ptr(tiles), ptr(n_tiles), ptr(wops), ptr(bn1),
                                        ptr(bns[1].weight.detach()), ptr(s2), ptr(zstar), ptr(arg), ptr(z2), V, st),
                  "dva_chain3_stats2")
        bn2 = _chain_bn(s2, V, bns[1], training)
        pooled = torch.empty((N, D), dtype=torch.float32, device=dev)
        check(lib.dva_chain_pooled(ptr(zstar), ptr(bn2), ptr(csr_idx), ptr(pooled), N, st), "dva_chain_pooled")
        t_add, set_saved = _set_branch_forward(e_map, pooled, csr_idx, training, zstats, prec3=True)
        # ---- layer 5: z5 = W5a act(BN2(z2)) + u[point] (+ statistics); z5 stays
        s5, s6 = zstats(), zstats()
        z5 = torch.empty((V, D), dtype=torch.float32, device=dev)
        with ops._timed("chain3_stats5", V * (4 + 2 * ROW) + N * 128):
            check(lib.dva_chain3_stats(5, ptr(z2), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops), ptr(bn2),
                                       ptr(z5), ptr(s5), V, N, st), "dva_chain3_stats")
        bn5 = _chain_bn(s5, V, bns[2], training)
        if training:
            with ops._timed("chain3_stats6", V * ROW):
                check(lib.dva_chain3_stats(6, ptr(z5), None, None, ptr(tiles), ptr(n_tiles), ptr(wops), ptr(bn5),
                                           None, ptr(s6), V, N, st), "dva_chain3_stats")
        bn6 = _chain_bn(s6, V, bns[3], training)
        scores = torch.empty((V, 4), dtype=torch.float32, device=dev)
        with ops._timed("chain3_scores", V * (ROW + 16)):
            check(lib.dva_chain3_scores(ptr(z5), ptr(tiles), ptr(n_tiles), ptr(wops), ptr(bn5), ptr(bn6), ptr(bs), G,
                                        ptr(scores), V, st), "dva_chain3_scores")
        need_bwd = any(ctx.needs_input_grad)
        if need_bwd:
            ctx.save_for_backward(x_map, csr_idx, vp, tiles, n_tiles, wops, t_add, zstar, arg, mom, bn1, bn2, bn5,
                                  bn6, W1, z2, z5)
        ctx.shim = shim
        ctx.set_saved = set_saved if need_bwd else None
        ctx.training = training
        ctx.G = G
        return scores if G == 4 else scores[:, :G].contiguous()

    @staticmethod
    def backward(ctx, dscores):
        lib = _lib.load()
        if ctx.set_saved is None:
            raise RuntimeError("the recompute chain's backward ran twice on the same graph: its per-step workspaces "
                               "are released after the first backward (retain_graph is not supported on this path)")
        (x_map, csr_idx, vp, tiles, n_tiles, wops, t_add, zstar, arg, mom, bn1, bn2, bn5, bn6, W1, z2,
         z5) = ctx.saved_tensors
        G, training = ctx.G, ctx.training
        dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
        st = stream_of(x_map)
        dc = dscores.contiguous().float()
        if G < 4:
            dc = torch.nn.functional.pad(dc, (0, 4 - G))
        arena = Arena(dev)
        m_rows = float(max(V, 1))
        zpool = iter(ops.zeros_small((10, 2 * D), torch.float64, dev))

        def zstats():
            return next(zpool)

        def consts(stats, bn, hat=True, out=True):
            return bn_bwd_consts(lib, arena, stats, bn, m_rows, training, st, hat, out)

        # ---- score layer: dWs, dbs, statistics of the BatchNorm-6 backward
        s6 = zstats()
        dWs, dbs = arena.take(G, D), arena.take(G)
        with ops._timed("chain3_score_stats", V * (ROW + 16)):
            check(lib.dva_chain3_score_stats(ptr(z5), ptr(tiles), ptr(n_tiles), ptr(wops), ptr(bn5), ptr(bn6),
                                             ptr(dc), ptr(s6), ptr(dWs), ptr(dbs), G, V, st),
                  "dva_chain3_score_stats")

        def layer(stage, x, zrows, vp_, u, bn_lo, bn_hi, sm, dc_, arg_, dpooled_, da_in, da_out, dW, du, P, stats,
                  name, nbytes):
            with ops._timed(name, nbytes):
                check(lib.dva_chain3_bwd_layer(stage, ptr(x), ptr(zrows), ptr(vp_), ptr(u), ptr(tiles), ptr(n_tiles),
                                               ptr(wops), ptr(bn_lo), ptr(bn_hi), ptr(sm), ptr(dc_), ptr(arg_),
                                               ptr(dpooled_), ptr(da_in), ptr(da_out), ptr(dW), ptr(du), ptr(P),
                                               ptr(stats), V, N, st), "dva_chain3_bwd_layer")

        sm6, g6, b6 = consts(s6, bn6)
        dW6 = arena.take(D, D)
        s5 = zstats()
        da5 = torch.empty((V, D), dtype=torch.float32, device=dev)
        layer(6, None, z5, None, None, bn5, bn6, sm6, dc, None, None, None, da5, dW6, None, None, s5,
              "chain3_bwd_l6", V * (2 * ROW + 16))
        del z5, dc
        sm5, g5, b5 = consts(s5, bn5)
        dW5 = arena.take(D, 2 * D)
        du = torch.zeros((N, D), dtype=torch.float32, device=dev)
        s2 = zstats()
        da2 = torch.empty((V, D), dtype=torch.float32, device=dev)
        layer(5, None, z2, vp, t_add, bn2, bn5, sm5, None, None, None, da5, da2, dW5, du, None, s2,
              "chain3_bwd_l5", V * (4 + 3 * ROW) + N * 256)
        del da5
        # ---- per-point set branch
        dpooled, d_set = _set_branch_backward(ctx.set_saved, du, dW5, training, zstats, arena)
        consts(s2, bn2, out=False)            # view part; the per-point part below is accumulated in z_hat directly
        dpooled_dy = torch.empty((N, D), dtype=torch.float32, device=dev)     # leaky'(y*) dpooled: what stage 2 routes
        check(lib.dva_chain_route_stats(ptr(zstar), ptr(dpooled), ptr(bn2), ptr(csr_idx), ptr(s2), ptr(dpooled_dy), N,
                                        st), "dva_chain_route_stats")
        sm2, g2, b2 = consts(s2, bn2, hat=False)
        dW2, P = arena.take(D, D), arena.take(D, 20)       # P = sum dy1 [x | 0 | 1]^T
        s1 = zstats()
        layer(2, x_map, z2, vp, None, bn1, bn2, sm2, None, arg, dpooled_dy, da2, None, dW2, None, P, None,
              "chain3_bwd_l2", V * (36 + 2 * ROW) + N * 256)
        del da2, z2
        check(lib.dva_chain_stats1(ptr(P), ptr(W1), 1, ptr(s1), st), "dva_chain_stats1")    # layer 1 is linear in x_map
        sm1, g1, b1 = consts(s1, bn1)
        dW1 = arena.take(D, 8)
        check(lib.dva_chain_dw1(ptr(P), ptr(mom), ptr(W1), 1, ptr(bn1), ptr(sm1), ptr(dW1), st), "dva_chain_dw1")
        ctx.set_saved = None
        grads = [dW1, g1, b1, dW2, g2, b2, dW5, g5, b5, dW6, g6, b6, dWs, dbs, None, None] + d_set
        return (None, None, None) + tuple(grads)


def chain_scores(e_map, linear, x_map, csr_idx):
    """``linear(e_map(x_map, csr_idx))`` fp32 [V, G <= 4]."""
    csr_idx = ops._check_ptr(csr_idx)
    shim = SimpleNamespace(E_map=e_map, E_score=linear, G=None)
    return _ChainScores.apply(x_map, csr_idx, shim, *chain_params(shim))
