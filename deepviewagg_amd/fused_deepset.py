"""Fused DeepSetFeat (+ trailing Linear) over the views: one autograd.Function driving the row-streaming
HIP kernels of ``csrc/deepset.hip`` (C ABI ``dva_deepset_*``), forward and hand-written backward.

Computes exactly ``linear(E_map(x_map, csr_idx))`` of the reference
(modules/multimodal/pooling.py:658-669 DeepSetFeat.forward followed by E_score :282 / K :484) for the
shipped configuration: ``DeepSetFeat(d_in=8, d_out=32, pool='max', fusion='concatenation',
use_num=*)``.  Train-mode BatchNorm uses batch statistics (accumulated in fp64 by the kernels) and
updates the running statistics like nn.BatchNorm1d; eval mode uses the running statistics.

The set branch (``mlp_set`` on the N points) goes through the same layer kernels with raw (un-normalised)
input; the set-size column of ``use_num`` is a rank-1 per-row addend.
"""
import torch
import torch.nn.functional as F

from . import _lib, ops
from ._lib import check, ptr, require_device, stream_of

D = 32
# layer-kernel generation: 0 = fp32 matrix cores, 1 = VALU + LDS broadcast (tests flip this)
ALGO = 0
# bf16 storage only: layer outputs that are cheap to recompute are not stored / not re-read -- a4 is never
# materialised, and the backward layer passes rebuild a_L from the layer input instead of reading it (one
# [V, 32] tensor less per pass for one more product on the matrix cores).  Measured on MI355X (S1/F-S):
# 8.6 GB less HBM traffic per step but 22.2 instead of 20.7 ms/step -- the passes are register / VALU limited
# once the extra product and its BatchNorm are added (spills in the x_map variant) -- so it is OFF by default
# and kept for parts with a lower bandwidth-to-VALU ratio.  Tests cover both settings.
RECOMPUTE = False
# storage type of the [V, 32] activation / gradient tensors between the kernels.  None = auto: bf16 inside
# torch.autocast(bfloat16) (what the reference's autocast keeps for these tensors; arithmetic, statistics
# and parameters stay fp32 here), fp32 otherwise.  Tests pin it.
ACT_DTYPE = None


def _act_dtype():
    if ACT_DTYPE is not None:
        return ACT_DTYPE
    if ALGO == 0 and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16:
        return torch.bfloat16
    return torch.float32


def _bn_of(block):
    return block[1].batch_norm


def applicable(e_map, linear, x_map):
    """Can ``linear(e_map(x_map))`` take the fused path?"""
    from .modules.multimodal import pooling as P
    if not isinstance(e_map, P.DeepSetFeat) or not isinstance(linear, torch.nn.Linear):
        return False
    if e_map.d_out != D or e_map.d_in != 8 or e_map.pool != ['max'] or e_map.fusion != 'concatenation':
        return False
    if linear.in_features != D or linear.out_features > 32 or linear.bias is None:
        return False
    if not (x_map.is_cuda and x_map.dtype == torch.float32 and x_map.dim() == 2 and x_map.shape[1] == 8):
        return False
    if x_map.requires_grad and torch.is_grad_enabled():
        return False  # gradient w.r.t. the raw mapping features is only produced by the generic path
    for mlp in (e_map.mlp_elt_1, e_map.mlp_set, e_map.mlp_elt_2):
        for block in mlp:
            bn = _bn_of(block)
            if block[0].bias is not None or not bn.affine or not bn.track_running_stats \
                    or bn.momentum is None or getattr(block[2], 'negative_slope', None) != 0.2:
                return False
            if bn.training != e_map.training:
                return False  # individually frozen BatchNorm layers: the generic path honours bn.training
            # the kernels read the parameters and running statistics through raw fp32 pointers
            if any(t.dtype != torch.float32 for t in (block[0].weight, bn.weight, bn.bias, bn.running_mean,
                                                      bn.running_var)):
                return False
    if linear.weight.dtype != torch.float32 or linear.bias.dtype != torch.float32:
        return False
    return True


def _bn_consts(stats, m, bn, training):
    """[4, 32] fp32 = mean | invstd | gamma | beta from batch statistics (training) or running stats;
    updates the running statistics in training (nn.BatchNorm1d semantics).  One kernel launch."""
    lib = _lib.load()
    out = torch.empty((4, D), dtype=torch.float32, device=stats.device)
    check(lib.dva_bn_finalize(ptr(stats), float(max(m, 1)), ptr(bn.running_mean), ptr(bn.running_var),
                              ptr(bn.num_batches_tracked), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                              float(bn.momentum), float(bn.eps), 1 if training else 0, D, ptr(out),
                              stream_of(stats)), "dva_bn_finalize")
    return out


class _DeepSetLinear(torch.autograd.Function):
    """params: Wa, g1, b1, Wb, g2, b2, Wc, g3, b3, Wd, g4, b4, Ws, bs, then the mlp_set parameters."""

    @staticmethod
    def forward(ctx, x_map, csr_idx, e_map, linear, *params):
        lib = _lib.load()
        require_device(x_map, csr_idx)
        x_map = x_map.contiguous()
        dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
        st = stream_of(x_map)
        training = e_map.training
        act = _act_dtype()
        AC, F32C = (_lib.DVA_BF16 if act == torch.bfloat16 else _lib.DVA_F32), _lib.DVA_F32
        RB = D * (2 if act == torch.bfloat16 else 4)      # bytes of one stored activation row
        Wa = e_map.mlp_elt_1[0][0].weight.detach().contiguous()
        Wb = e_map.mlp_elt_1[1][0].weight.detach().contiguous()
        Wc = e_map.mlp_elt_2[0][0].weight.detach()
        WcA = Wc[:, :D].contiguous()
        Wd = e_map.mlp_elt_2[1][0].weight.detach().contiguous()
        Ws, bs = linear.weight.detach().contiguous(), linear.bias.detach().contiguous()
        G = Ws.shape[0]
        bns = [_bn_of(e_map.mlp_elt_1[0]), _bn_of(e_map.mlp_elt_1[1]),
               _bn_of(e_map.mlp_elt_2[0]), _bn_of(e_map.mlp_elt_2[1])]

        zpool = iter(torch.zeros((8, 2 * D), dtype=torch.float64, device=dev))

        def zstats():
            return next(zpool)

        # ---- elt MLP 1: x_map -> a1 -> a2
        s1 = zstats()
        if training:
            with ops._timed("deepset_fwd_first_stats", V * 32):
                check(lib.dva_deepset_fwd_first(ptr(x_map), ptr(Wa), None, None, None, ptr(s1), V, 8, 1, ALGO, AC, st),
                      "dva_deepset_fwd_first")
        bn1 = _bn_consts(s1, V, bns[0], training)
        a2 = torch.empty((V, D), dtype=act, device=dev)
        s2 = zstats()
        with ops._timed("deepset_fwd_first", V * (32 + RB)):
            check(lib.dva_deepset_fwd_first(ptr(x_map), ptr(Wa), ptr(bn1), ptr(Wb), ptr(a2), ptr(s2), V, 8, 0, ALGO, AC, st),
                  "dva_deepset_fwd_first")
        bn2 = _bn_consts(s2, V, bns[1], training)
        # ---- set branch: max over views, set MLP on the N points, WcB product (PyTorch, N rows)
        pooled = torch.empty((N, D), dtype=torch.float32, device=dev)
        arg = torch.empty((N, D), dtype=torch.int32, device=dev)
        with ops._timed("deepset_segmax", V * RB + N * (256 + 8)):
            check(lib.dva_deepset_segmax(ptr(a2), ptr(bn2), ptr(csr_idx), ptr(pooled), ptr(arg), N, V, AC, st),
                  "dva_deepset_segmax")
        # set MLP on the N points with the same layer kernels (raw input, the set-size column of
        # use_num enters as a rank-1 per-row addend), then the WcB half of the concatenation layer
        mlp_set = e_map.mlp_set
        Wsa_full = mlp_set[0][0].weight.detach()
        WsaP = Wsa_full[:, :D].contiguous()
        Wsb = mlp_set[1][0].weight.detach().contiguous()
        WcB = Wc[:, D:].contiguous()
        set_bns = [_bn_of(mlp_set[0]), _bn_of(mlp_set[1])]
        num = add1 = ident_idx = None
        if e_map.use_num:
            sizes = csr_idx[1:] - csr_idx[:-1]
            num = torch.sqrt(1 / (sizes + 1e-3)).float()
            add1 = (num.view(-1, 1) * Wsa_full[:, D].view(1, -1)).contiguous()
            ident_idx = torch.arange(N, dtype=torch.int32, device=dev)
        u1, su1 = torch.empty((N, D), dtype=torch.float32, device=dev), zstats()
        check(lib.dva_deepset_fwd_layer(ptr(pooled), None, ptr(WsaP), ptr(add1), ptr(ident_idx), ptr(u1), ptr(su1),
                                        N, 0, F32C, st), "dva_deepset_fwd_layer")
        bns1 = _bn_consts(su1, N, set_bns[0], training)
        u2, su2 = torch.empty((N, D), dtype=torch.float32, device=dev), zstats()
        check(lib.dva_deepset_fwd_layer(ptr(u1), ptr(bns1), ptr(Wsb), None, None, ptr(u2), ptr(su2), N, 0, F32C, st),
              "dva_deepset_fwd_layer")
        bns2 = _bn_consts(su2, N, set_bns[1], training)
        t_add, su3 = torch.empty((N, D), dtype=torch.float32, device=dev), zstats()
        check(lib.dva_deepset_fwd_layer(ptr(u2), ptr(bns2), ptr(WcB), None, None, ptr(t_add), ptr(su3), N, 0, F32C, st),
              "dva_deepset_fwd_layer")
        vp = torch.empty(V, dtype=torch.int32, device=dev)
        check(lib.dva_csr_expand(ptr(csr_idx), N, ptr(vp), st), "dva_csr_expand")
        # ---- elt MLP 2: cat(h1, set[p]) -> a3 -> a4
        t_det = t_add
        a3 = torch.empty((V, D), dtype=act, device=dev)
        s3 = zstats()
        with ops._timed("deepset_fwd_layer_add", V * (2 * RB + 4) + N * 128):
            check(lib.dva_deepset_fwd_layer(ptr(a2), ptr(bn2), ptr(WcA), ptr(t_det), ptr(vp), ptr(a3), ptr(s3), V, ALGO,
                                            AC, st),
                  "dva_deepset_fwd_layer")
        bn3 = _bn_consts(s3, V, bns[2], training)
        recompute = RECOMPUTE and act == torch.bfloat16 and ALGO == 0
        s4 = zstats()
        if recompute:
            # statistics of a4 only; the scores pass recomputes a4 from a3 in registers
            a4 = None
            with ops._timed("deepset_fwd_layer_stats", V * RB):
                check(lib.dva_deepset_fwd_layer(ptr(a3), ptr(bn3), ptr(Wd), None, None, None, ptr(s4), V, ALGO, AC, st),
                      "dva_deepset_fwd_layer")
        else:
            a4 = torch.empty((V, D), dtype=act, device=dev)
            with ops._timed("deepset_fwd_layer", V * 2 * RB):
                check(lib.dva_deepset_fwd_layer(ptr(a3), ptr(bn3), ptr(Wd), None, None, ptr(a4), ptr(s4), V, ALGO, AC, st),
                      "dva_deepset_fwd_layer")
        bn4 = _bn_consts(s4, V, bns[3], training)
        # ---- trailing Linear (E_score / K)
        out = torch.empty((V, G), dtype=torch.float32, device=dev)
        with ops._timed("deepset_fwd_score", V * (RB + 4 * G)):
            if recompute:
                check(lib.dva_deepset_fwd_score(ptr(a3), ptr(bn4), ptr(Ws), ptr(bs), ptr(out), V, G, ptr(bn3), ptr(Wd),
                                                ALGO, AC, st), "dva_deepset_fwd_score")
            else:
                check(lib.dva_deepset_fwd_score(ptr(a4), ptr(bn4), ptr(Ws), ptr(bs), ptr(out), V, G, None, None,
                                                ALGO, AC, st), "dva_deepset_fwd_score")

        ctx.save_for_backward(x_map, csr_idx, vp, a2, a3, a4 if a4 is not None else a3, arg, bn1, bn2, bn3, bn4,
                              Wa, Wb, WcA, Wd, Ws)
        ctx.recompute = recompute
        ctx.set_branch = (pooled, u1, u2, t_add, bns1, bns2, WsaP, Wsb, WcB, num, ident_idx)
        ctx.modules = (e_map, linear)
        ctx.training = training
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x_map, csr_idx, vp, a2, a3, a4, arg, bn1, bn2, bn3, bn4, Wa, Wb, WcA, Wd, Ws = ctx.saved_tensors
        e_map, linear = ctx.modules
        pooled, u1, u2, t_add, bns1, bns2, WsaP, Wsb, WcB, num, ident_idx = ctx.set_branch
        dev, V, N, G = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1, Ws.shape[0]
        st = stream_of(x_map)
        dout = dout.contiguous().float()
        m = float(max(V, 1))
        act = ctx.act
        AC, F32C = (_lib.DVA_BF16 if act == torch.bfloat16 else _lib.DVA_F32), _lib.DVA_F32
        RB = D * (2 if act == torch.bfloat16 else 4)

        zpool = iter(torch.zeros((8, 2 * D), dtype=torch.float64, device=dev))

        def zstats():
            return next(zpool)

        def sm_of(stats, rows=m):
            # S1/M | S2/M of the batch-statistics BN backward; zero with running statistics (eval)
            if not ctx.training:
                return torch.zeros(2 * D, dtype=torch.float32, device=dev)
            out = torch.empty(2 * D, dtype=torch.float32, device=dev)
            check(lib.dva_scale_f64(ptr(stats), 1.0 / rows, ptr(out), 2 * D, st), "dva_scale_f64")
            return out

        def buf(rows=V, dtype=None):
            return torch.empty((rows, D), dtype=act if dtype is None else dtype, device=dev)

        # score layer
        dz4, s4 = buf(), zstats()
        dWs = torch.zeros_like(Ws)
        dbs = torch.zeros(G, dtype=torch.float32, device=dev)
        rc = ctx.recompute          # a4 was never stored (the saved slot holds a3); a_L rebuilt from the layer input
        with ops._timed("deepset_bwd_score", V * (2 * RB + 4 * G)):
            check(lib.dva_deepset_bwd_score(ptr(dout), ptr(a3 if rc else a4), ptr(bn4), ptr(Ws), ptr(dz4), ptr(dWs),
                                            ptr(dbs), ptr(s4), V, G, ptr(bn3) if rc else None,
                                            ptr(Wd) if rc else None, ALGO, AC, st), "dva_deepset_bwd_score")
        # Wd layer (a3 -> a4)
        dz3, s3, dWd = buf(), zstats(), torch.zeros_like(Wd)
        sm4 = sm_of(s4)
        with ops._timed("deepset_bwd_layer", V * RB * (3 if rc else 4)):
            check(lib.dva_deepset_bwd_layer(ptr(dz4), None if rc else ptr(a4), ptr(bn4), ptr(sm4), ptr(Wd), ptr(a3),
                                            None, ptr(bn3), ptr(dz3), ptr(dWd), ptr(s3), None, None, None, None, V,
                                            0, 0, ALGO, AC, st), "dva_deepset_bwd_layer")
        del dz4
        # Wc layer (cat(h1, set) -> a3): raw dx on the h1 half, per-point sum on the set half
        dcat, dWcA = buf(), torch.zeros_like(WcA)
        dt = torch.zeros((N, D), dtype=torch.float32, device=dev)
        sm3 = sm_of(s3)
        with ops._timed("deepset_bwd_layer_cat", V * (RB * (3 if rc else 4) + 4) + N * 128):
            check(lib.dva_deepset_bwd_layer(ptr(dz3), None if rc else ptr(a3), ptr(bn3), ptr(sm3), ptr(WcA), ptr(a2),
                                            None, ptr(bn2), ptr(dcat), ptr(dWcA), None, ptr(dt), ptr(vp), None,
                                            ptr(t_add) if rc else None, V, 0, 1, ALGO, AC, st),
                  "dva_deepset_bwd_layer")
        del dz3
        # set branch backward with the same layer kernels over the N points
        n_rows = float(max(N, 1))
        ident_bn = torch.zeros(4 * D, dtype=torch.float32, device=dev)   # (mean 0 | invstd 1 | gamma 1 | beta 0),
        ident_bn[D:3 * D] = 1.0                                          # built on the device: no host copy
        zero_sm = torch.zeros(2 * D, dtype=torch.float32, device=dev)
        dWcB = torch.zeros_like(WcB)
        dzs2, ss2 = buf(N, torch.float32), zstats()
        check(lib.dva_deepset_bwd_layer(ptr(dt), ptr(t_add), ptr(ident_bn), ptr(zero_sm), ptr(WcB), ptr(u2), None,
                                        ptr(bns2), ptr(dzs2), ptr(dWcB), ptr(ss2), None, None, None, None, N, 0, 0, 0, F32C, st),
              "dva_deepset_bwd_layer")
        dWsb = torch.zeros_like(Wsb)
        dzs1, ss1 = buf(N, torch.float32), zstats()
        sms2 = sm_of(ss2, n_rows)
        check(lib.dva_deepset_bwd_layer(ptr(dzs2), ptr(u2), ptr(bns2), ptr(sms2), ptr(Wsb), ptr(u1), None,
                                        ptr(bns1), ptr(dzs1), ptr(dWsb), ptr(ss1), None, None, None, None, N, 0, 0, 0, F32C, st),
              "dva_deepset_bwd_layer")
        dWsaP = torch.zeros_like(WsaP)
        dpooled = buf(N, torch.float32)
        da1 = torch.zeros((N, D), dtype=torch.float32, device=dev) if num is not None else None
        sms1 = sm_of(ss1, n_rows)
        check(lib.dva_deepset_bwd_layer(ptr(dzs1), ptr(u1), ptr(bns1), ptr(sms1), ptr(WsaP), ptr(pooled), None,
                                        None, ptr(dpooled), ptr(dWsaP), None, ptr(da1), ptr(ident_idx), None, None, N, 0,
                                        1, 0, F32C, st), "dva_deepset_bwd_layer")
        if num is not None:
            dWsa = torch.cat([dWsaP, (da1 * num.view(-1, 1)).sum(0).view(-1, 1)], dim=1)
        else:
            dWsa = dWsaP
        dWc = torch.cat([dWcA, dWcB], dim=1)
        d_set = [dWsa, ss1[D:].float(), ss1[:D].float(), dWsb, ss2[D:].float(), ss2[:D].float()]
        # join the max path, BN2 backward statistics
        dz2, s2 = buf(), zstats()
        with ops._timed("deepset_bwd_max", V * (RB * 3 + 4) + N * 256):
            check(lib.dva_deepset_bwd_max(ptr(dcat), ptr(a2), ptr(bn2), ptr(arg), ptr(dpooled), ptr(vp), ptr(dz2),
                                          ptr(s2), V, ALGO, AC, st), "dva_deepset_bwd_max")
        del dcat
        # Wb layer (a1 -> a2), a1 recomputed from x_map
        s1, dWb = zstats(), torch.zeros_like(Wb)
        sm2 = sm_of(s2)
        if act == torch.bfloat16 and ALGO == 0:
            # first layer folded in: the pass also accumulates P | Q | SX (BN1-backward is linear in the
            # statistics it produces), dz1 is never written and there is no bwd_first pass
            first = torch.zeros(520, dtype=torch.float32, device=dev)
            with ops._timed("deepset_bwd_layer_xmap_first", V * (RB * (1 if rc else 2) + 32)):
                check(lib.dva_deepset_bwd_layer(ptr(dz2), None if rc else ptr(a2), ptr(bn2), ptr(sm2), ptr(Wb),
                                                ptr(x_map), ptr(Wa), ptr(bn1), None, ptr(dWb), ptr(s1), None, None,
                                                ptr(first), None, V, 1, 0, ALGO, AC, st), "dva_deepset_bwd_layer")
            del dz2
            sm1 = sm_of(s1)
            P, Q, SX = first[:256].view(D, 8), first[256:512].view(D, 8), first[512:]
            gsc = (bn1[2] * bn1[1]).view(D, 1)
            dWa = gsc * (P - sm1[:D].view(D, 1) * SX.view(1, 8) - sm1[D:].view(D, 1) * Q)
        else:
            dz1 = buf()
            with ops._timed("deepset_bwd_layer_xmap", V * (RB * 3 + 32)):
                check(lib.dva_deepset_bwd_layer(ptr(dz2), ptr(a2), ptr(bn2), ptr(sm2), ptr(Wb), ptr(x_map), ptr(Wa),
                                                ptr(bn1), ptr(dz1), ptr(dWb), ptr(s1), None, None, None, None, V, 1, 0,
                                                ALGO, AC, st), "dva_deepset_bwd_layer")
            del dz2
            dWa = torch.zeros_like(Wa)
            sm1 = sm_of(s1)
            with ops._timed("deepset_bwd_first", V * (RB + 32)):
                check(lib.dva_deepset_bwd_first(ptr(dz1), ptr(x_map), ptr(Wa), ptr(bn1), ptr(sm1), ptr(dWa), V, 8, AC,
                                                st), "dva_deepset_bwd_first")

        def gb(stats):  # d gamma = S2, d beta = S1
            return stats[D:].float(), stats[:D].float()
        g1, b1 = gb(s1)
        g2, b2 = gb(s2)
        g3, b3 = gb(s3)
        g4, b4 = gb(s4)
        grads = [dWa, g1, b1, dWb, g2, b2, dWc, g3, b3, dWd, g4, b4, dWs, dbs] + d_set
        ctx.set_branch = None
        return (None, None, None, None) + tuple(grads)


def deepset_linear(e_map, linear, x_map, csr_idx):
    """``linear(e_map(x_map, csr_idx))`` through the fused kernels (check ``applicable`` first)."""
    blocks = (e_map.mlp_elt_1[0], e_map.mlp_elt_1[1], e_map.mlp_elt_2[0], e_map.mlp_elt_2[1])
    params = []
    for blk in blocks:
        bn = _bn_of(blk)
        params += [blk[0].weight, bn.weight, bn.bias]
    params += [linear.weight, linear.bias]
    params += list(e_map.mlp_set.parameters())
    csr_idx = ops._check_ptr(csr_idx)
    return _DeepSetLinear.apply(x_map, csr_idx, e_map, linear, *params)
