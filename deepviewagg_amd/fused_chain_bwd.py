"""Backward of fused_chain._ChainPool: four view passes (attention, layer 6, layer 5, layer 2), each re-evaluating
the DeepSetFeat chain from x_map, separated by the BatchNorm-backward statistics; the per-point set branch in
between runs through the fp32 layer kernels (fused_chain._set_branch_backward)."""
import torch

from . import _lib, ops
from ._lib import check, ptr, stream_of
from .fused_deepset import D


def backward(ctx, gout):
    from .fused_chain import _set_branch_backward
    lib = _lib.load()
    (rows, row_idx, x_map, csr_idx, vp, tiles, n_tiles, wops, t_add, zstar, arg, mom,
     bn1, bn2, bn5, bn6, out) = ctx.saved_tensors
    module, training = ctx.module, ctx.training
    scaling, eps = ctx.meta
    e_map, e_score, gate = module.E_map, module.E_score, module.G
    dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
    R, C = rows.shape
    G = e_score.weight.shape[0]
    st = stream_of(x_map)
    gout = gout.contiguous().to(torch.bfloat16)
    m_rows = float(max(V, 1))
    bs = e_score.bias.detach().contiguous()
    gw = gate.weight.detach().reshape(-1).float().contiguous() if gate is not None else None
    gb = gate.bias.detach().reshape(-1).float().contiguous() if gate is not None else None

    zpool = iter(torch.zeros((10, 2 * D), dtype=torch.float64, device=dev))

    def zstats():
        return next(zpool)

    def to_hat(stats, bn):
        """The kernels accumulate S1 = sum dy and sum dy z (raw layer output): S2 = sum dy z_hat =
        invstd (sum dy z - mean S1), in place."""
        stats[D:] = bn[1].double() * (stats[D:] - bn[0].double() * stats[:D])

    def sm_of(stats):
        if not training:
            return torch.zeros(2 * D, dtype=torch.float32, device=dev)
        o = torch.empty(2 * D, dtype=torch.float32, device=dev)
        check(lib.dva_scale_f64(ptr(stats), 1.0 / m_rows, ptr(o), 2 * D, st), "dva_scale_f64")
        return o

    # ---- attention + gate backward: score gradients, view records, S6
    dc = torch.empty((V, 4), dtype=torch.float32, device=dev)
    rec = torch.empty((V, 8), dtype=torch.float32, device=dev)
    s6 = zstats()
    gwb = torch.zeros(2 * G, dtype=torch.float32, device=dev) if gate is not None else None
    with ops._timed("chain_attn_bwd", V * (C * 2 + 32 + 8 + 16 + 32) + N * (2 * C * 2 + 128 + 8)):
        check(lib.dva_chain_attn_bwd(ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                     ptr(bn1), ptr(bn2), ptr(bn5), ptr(bn6), ptr(bs), ptr(rows), ptr(row_idx),
                                     ptr(csr_idx), ptr(gw), ptr(gb), ptr(gout), ptr(out), ptr(dc), ptr(rec),
                                     ptr(s6), ptr(gwb), N, V, R, C, G, scaling, eps, st), "dva_chain_attn_bwd")
    # ---- rows gradient: segmented reduction over the row plan (deterministic, no atomics)
    grows = None
    if ctx.needs_input_grad[0]:
        plan = ctx.plan if ctx.plan is not None else ops.row_plan(row_idx, R, with_counts=False)[0]
        perm, row_ptr = plan
        grows = torch.empty((R, C), dtype=torch.float32, device=dev)
        with ops._timed("view_gather_rows_grad", V * (4 + 32 + C * 2) + R * (C * 4 + 4)):
            check(lib.dva_view_gather_rows_grad(ptr(gout), None, None, None, ptr(perm), ptr(row_ptr), ptr(rec), 8,
                                                ptr(grows), R, V, C, G, _lib.DVA_BF16, st),
                  "dva_view_gather_rows_grad")
        grows = grows.to(rows.dtype)
    del rec

    def layer(stage, sm2, sm5, sm6, arg_, dpooled_, da_in, da_out, dW, dWs, dbs, du, P, stats, name, nbytes):
        with ops._timed(name, nbytes):
            check(lib.dva_chain_bwd_layer(stage, ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                          ptr(bn1), ptr(bn2), ptr(bn5), ptr(bn6), ptr(sm2), ptr(sm5), ptr(sm6),
                                          ptr(dc), ptr(arg_), ptr(dpooled_), ptr(da_in), ptr(da_out), ptr(dW),
                                          ptr(dWs), ptr(dbs), ptr(du), ptr(P), ptr(stats), G, V, N, st),
                  "dva_chain_bwd_layer")

    # per view: x_map 32 + view->point 4 (+ score gradients 16) + the 64-byte gradient row handed between the passes
    to_hat(s6, bn6)
    sm6 = sm_of(s6)
    dW6 = torch.zeros((D, D), dtype=torch.float32, device=dev)
    dWs = torch.zeros((G, D), dtype=torch.float32, device=dev)
    dbs = torch.zeros(G, dtype=torch.float32, device=dev)
    s5 = zstats()
    da5 = torch.empty((V, D), dtype=torch.bfloat16, device=dev)
    layer(6, None, None, sm6, None, None, None, da5, dW6, dWs, dbs, None, None, s5, "chain_bwd_l6",
          V * (32 + 4 + 16 + 64) + N * 128)
    del dc
    to_hat(s5, bn5)
    sm5 = sm_of(s5)
    dW5 = torch.zeros((D, 2 * D), dtype=torch.float32, device=dev)
    du = torch.zeros((N, D), dtype=torch.float32, device=dev)
    s2 = zstats()
    da2 = torch.empty((V, D), dtype=torch.bfloat16, device=dev)
    dc = None
    layer(5, None, sm5, None, None, None, da5, da2, dW5, None, None, du, None, s2, "chain_bwd_l5",
          V * (32 + 4 + 64 + 64) + N * 256)
    del da5
    # ---- per-point set branch
    dpooled, d_set = _set_branch_backward(ctx.set_saved, du, dW5, training, zstats)
    to_hat(s2, bn2)            # view part; the per-point part below is accumulated in z_hat directly
    check(lib.dva_chain_route_stats(ptr(zstar), ptr(dpooled), ptr(bn2), ptr(csr_idx), ptr(s2), N, st),
          "dva_chain_route_stats")
    sm2 = sm_of(s2)
    dW2 = torch.zeros((D, D), dtype=torch.float32, device=dev)
    P = torch.zeros((D, 8), dtype=torch.float32, device=dev)
    s1 = zstats()
    layer(2, sm2, None, None, arg, dpooled, da2, None, dW2, None, None, None, P, s1, "chain_bwd_l2",
          V * (32 + 4 + 64) + N * 256)
    del da2
    to_hat(s1, bn1)
    sm1 = sm_of(s1)
    # ---- first layer: BatchNorm-1 backward is linear in its statistics and z1 = W1 x is linear in x, so
    #      dW1 = G1 (P - (S1/M) SX^T - (S2/M) . Q) with Q = sum_v z1_hat x^T from the moments of x_map
    W1b = e_map.mlp_elt_1[0][0].weight.detach().to(torch.bfloat16).double()
    momd = mom
    SX = momd[:8]
    XX = torch.zeros((8, 8), dtype=torch.float64, device=dev)
    iu = torch.triu_indices(8, 8, device=dev)
    XX[iu[0], iu[1]] = momd[8:]
    XX = XX + XX.t() - torch.diag(torch.diagonal(XX))
    mean1, inv1, gam1 = bn1[0].double(), bn1[1].double(), bn1[2].double()
    Q = inv1.view(D, 1) * (W1b @ XX - mean1.view(D, 1) * SX.view(1, 8))
    sm1d = sm1.double()
    dW1 = ((gam1 * inv1).view(D, 1) * (P.double() - sm1d[:D].view(D, 1) * SX.view(1, 8)
                                       - sm1d[D:].view(D, 1) * Q)).float()

    def gb_of(stats):   # d gamma = S2, d beta = S1
        return stats[D:].float(), stats[:D].float()
    g1, b1 = gb_of(s1)
    g2, b2 = gb_of(s2)
    g5, b5 = gb_of(s5)
    g6, b6 = gb_of(s6)
    if gate is not None:
        dgw, dgb = gwb[:G].reshape(gate.weight.shape), gwb[G:].reshape(gate.bias.shape)
    else:
        dgw = dgb = None
    grads = [dW1, g1, b1, dW2, g2, b2, dW5, g5, b5, dW6, g6, b6, dWs, dbs, dgw, dgb] + d_set
    ctx.set_saved = None
    return (grows, None, None, None, None, None, None, None) + tuple(grads)
