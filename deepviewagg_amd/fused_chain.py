"""GroupBimodalCSRPool on a lazily gathered bf16 value map through the recompute chain
(``csrc/chain_fwd.hip`` / ``csrc/chain_bwd.hip``, C ABI ``dva_chain_*``).

Computes ``gate * sum_v softmax_v(E_score(E_map(x_map))) * rows[row_idx[v]]`` of the reference
(modules/multimodal/pooling.py:263-315 with map_encoder = DeepSetFeat :658-669) without any [V, .]
activation tensor: every kernel re-evaluates the per-view DeepSetFeat chain from the 32-byte mapping features
(bf16 matrix cores, layers chained in registers).  The only view-sized reads are ``x_map``, the view -> point
index and the row index; the only view-sized writes of a training step are the score gradients [V, G], the
16-byte view records of the rows gradient and one bf16 [V, 32] gradient row handed between two backward passes.
Train-mode BatchNorm keeps one statistics pass per layer.

Selected by ``pooling.GroupBimodalCSRPool`` inside ``torch.autocast(bfloat16)`` (or with ``FORCE = True``)
when ``applicable`` holds; everything else takes the fp32 kernels of ``fused_deepset`` / ``ops``.
The per-point set branch (``mlp_set`` on N rows) runs on the same kind of kernels (``csrc/chain_set.hip``).
"""
import os

import torch

from . import _lib, ops, fused_deepset
from ._lib import check, ptr, require_device, stream_of
from .fused_deepset import D, _bn_of, _bn_consts

# None = auto (inside torch.autocast(bfloat16) only); True / False pin the choice (tests, bench A/B)
FORCE = None
VIEWS_PER_CHUNK = 512       # tile-table construction granularity (one lane walks one chunk)
OPS_BYTES = 18 * 64 * 16 + 7 * 64 * 32       # bf16 operand blocks + the fp32 copy of the forward operands


def enabled():
    if FORCE is not None:
        return FORCE
    return torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16


def applicable(module, x_mod, x_map, csr_idx=None):
    """Can ``module`` (a GroupBimodalCSRPool) pool ``x_mod`` (an ops.GatheredFeatures) on the chain?"""
    if not enabled() or module.use_mod or module.save_last:
        return False
    if not isinstance(x_mod, ops.GatheredFeatures) or x_mod.rows.dtype != torch.bfloat16:
        return False
    if not fused_deepset.applicable(module.E_map, module.E_score, x_map):        # incl. fp32 parameters / buffers
        return False
    if module.G is not None and any(t is not None and t.dtype != torch.float32 for t in (module.G.weight, module.G.bias)):
        return False
    C, G = module.out_mod, module.num_groups
    if C not in (32, 64, 128, 256, 512) or G not in (1, 2, 4) or C % G or (C // G) % 8:
        return False
    # 32-bit buffer addressing inside the kernels: x_map (32 B / view), value rows, the [V, 32] bf16 gradient row
    V, R = x_map.shape[0], x_mod.rows.shape[0]
    N = csr_idx.shape[0] - 1 if csr_idx is not None else 0
    return V * 64 < (1 << 32) - 16 and R * C * 2 < (1 << 32) - 16 and N * max(C * 2, 128) < (1 << 32) - 16


def pooled_output(csr_idx, N, C, dev):
    """The bf16 [N, C] tensor the view kernel pools into: the rows of points WITHOUT views are cleared here (exact
    zeros, pooling.py:870), the kernel writes every other row -- no fill of the whole tensor (round 5)."""
    lib = _lib.load()
    out = torch.empty((N, C), dtype=torch.bfloat16, device=dev)
    check(lib.dva_zero_unseen_rows(ptr(csr_idx), ptr(out), N, C * 2, stream_of(out)), "dva_zero_unseen_rows")
    return out


def build_tiles(csr_idx, V):
    """Tile table of a CSR pointer array: (tiles int32 [T_max, 2], n_tiles int32 [1]), no host sync."""
    lib = _lib.load()
    dev, N = csr_idx.device, csr_idx.shape[0] - 1
    st = stream_of(csr_idx)
    n_chunks = max(1, min(1 << 17, (V + VIEWS_PER_CHUNK - 1) // VIEWS_PER_CHUNK))
    step = (V + n_chunks - 1) // n_chunks if V > 0 else 1
    cp = torch.empty(n_chunks + 1, dtype=torch.int64, device=dev)
    counts = torch.empty(n_chunks, dtype=torch.int32, device=dev)
    offsets = torch.empty(n_chunks, dtype=torch.int64, device=dev)
    n_tiles = torch.empty(1, dtype=torch.int32, device=dev)
    t_max = min(N, V) + V // 32 + 1
    tiles = torch.empty((t_max, 2), dtype=torch.int32, device=dev)
    check(lib.dva_chain_tile_chunks(ptr(csr_idx), N, step, n_chunks, ptr(cp), st), "dva_chain_tile_chunks")
    check(lib.dva_chain_tile_count(ptr(csr_idx), ptr(cp), n_chunks, ptr(counts), st), "dva_chain_tile_count")
    check(lib.dva_chain_tile_offsets(ptr(counts), n_chunks, ptr(offsets), ptr(n_tiles), st),
          "dva_chain_tile_offsets")
    check(lib.dva_chain_tile_build(ptr(csr_idx), ptr(cp), n_chunks, ptr(offsets), ptr(tiles), st),
          "dva_chain_tile_build")
    return tiles, n_tiles


def _chain_bn(stats, m, bn, training, W=None, K=0, sum_a=None):
    """fp32 [5, 32] = mean | invstd | gamma | beta | shift of one chain layer (dva_chain_bn_consts): batch statistics
    (training; running statistics updated as nn.BatchNorm1d does) or running statistics; ``W`` / ``sum_a`` for a layer
    that the passes evaluate with BatchNorm folded into the weight operand (exact batch mean of the folded product)."""
    lib = _lib.load()
    out = torch.empty((5, D), dtype=torch.float32, device=stats.device)
    fold = training and W is not None and sum_a is not None
    check(lib.dva_chain_bn_consts(ptr(stats), float(max(m, 1)), ptr(bn.running_mean), ptr(bn.running_var),
                                  ptr(bn.num_batches_tracked), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                                  float(bn.momentum), float(bn.eps), 1 if training else 0,
                                  ptr(W) if fold else None, W.shape[1] if fold else 0, K if fold else 0,
                                  ptr(sum_a) if fold else None, ptr(out), stream_of(stats)), "dva_chain_bn_consts")
    return out


SET_OPS_BYTES = 24 * 64 * 16


def _set_fns(lib, prec3):
    if prec3:
        return lib.dva_chain3_set_prep, lib.dva_chain3_set_fwd, lib.dva_chain3_set_bwd
    return lib.dva_chain_set_prep, lib.dva_chain_set_fwd, lib.dva_chain_set_bwd


def _set_branch_forward(e_map, pooled, csr_idx, training, zstats, prec3=False):
    """mlp_set on the N points + the per-point half of the concatenation layer on the chain kernels of
    csrc/chain_set.hip (every pass re-evaluates the branch from ``pooled``; ``prec3``: their fp32 twins of
    csrc/chain_f32.hip).  Returns (t_add, saved)."""
    lib = _lib.load()
    set_prep, set_fwd, _ = _set_fns(lib, prec3)
    dev, N = pooled.device, pooled.shape[0]
    st = stream_of(pooled)
    mlp_set = e_map.mlp_set
    Wc = e_map.mlp_elt_2[0][0].weight.detach().contiguous()
    Wsa = mlp_set[0][0].weight.detach().contiguous()               # [32, 32 (+ 1 with use_num)]
    Wsb = mlp_set[1][0].weight.detach().contiguous()
    set_bns = [_bn_of(mlp_set[0]), _bn_of(mlp_set[1])]
    w33 = Wsa[:, D].contiguous() if e_map.use_num else None
    sops = torch.empty(SET_OPS_BYTES, dtype=torch.uint8, device=dev)
    check(set_prep(ptr(Wsa), Wsa.shape[1], ptr(Wsb), ptr(Wc), Wc.shape[1], ptr(sops), st), "dva_chain_set_prep")
    su1, su2 = zstats(), zstats()
    if training:
        check(set_fwd(1, ptr(pooled), ptr(csr_idx), ptr(w33), ptr(sops), None, None, None, ptr(su1), N, st),
              "dva_chain_set_fwd")
    bns1 = _bn_consts(su1, N, set_bns[0], training)
    if training:
        check(set_fwd(2, ptr(pooled), ptr(csr_idx), ptr(w33), ptr(sops), ptr(bns1), None, None, ptr(su2), N, st),
              "dva_chain_set_fwd")
    bns2 = _bn_consts(su2, N, set_bns[1], training)
    t_add = torch.empty((N, D), dtype=torch.float32, device=dev)
    check(set_fwd(3, ptr(pooled), ptr(csr_idx), ptr(w33), ptr(sops), ptr(bns1), ptr(bns2), ptr(t_add), None, N, st),
          "dva_chain_set_fwd")
    return t_add, (pooled, csr_idx, w33, sops, bns1, bns2, tuple(Wsa.shape), prec3)


def _set_branch_backward(saved, dt, dWc, training, zstats, arena):
    """Backward of _set_branch_forward: dt [N, 32] = gradient of t_add.  d Wc[:, 32:] is accumulated into ``dWc``
    [32, 64] in place.  Returns (dpooled, d_set) with d_set = gradients of list(mlp_set.parameters())."""
    from .fused_chain_bwd import bn_bwd_consts
    lib = _lib.load()
    pooled, csr_idx, w33, sops, bns1, bns2, wsa_shape, prec3 = saved
    set_bwd = _set_fns(lib, prec3)[2]
    dev, N = pooled.device, pooled.shape[0]
    st = stream_of(pooled)
    n_rows = float(max(N, 1))

    def call(stage, sm1, sm2, dpooled, dW, ld, dw33, stats):
        check(set_bwd(stage, ptr(pooled), ptr(csr_idx), ptr(w33), ptr(sops), ptr(bns1), ptr(bns2), ptr(sm1), ptr(sm2),
                      ptr(dt), ptr(dpooled), ptr(dW), ld, ptr(dw33), ptr(stats), N, st), "dva_chain_set_bwd")
    ss2, ss1 = zstats(), zstats()
    call(1, None, None, None, dWc[:, D:], dWc.shape[1], None, ss2)
    sms2, g2, b2 = bn_bwd_consts(lib, arena, ss2, bns2, n_rows, training, st)
    dWsb = arena.take(D, D)
    call(2, None, sms2, None, dWsb, D, None, ss1)
    sms1, g1, b1 = bn_bwd_consts(lib, arena, ss1, bns1, n_rows, training, st)
    dWsa = arena.take(*wsa_shape)
    dpooled = torch.empty((N, D), dtype=torch.float32, device=dev)
    call(3, sms1, sms2, dpooled, dWsa, wsa_shape[1], dWsa[:, D:] if w33 is not None else None, None)
    return dpooled, [dWsa, g1, b1, dWsb, g2, b2]


# DVA_CHAIN_A2=1 -- the stored-a2 hybrid (round 6, VERDICT r5 item 3: bytes for instructions): the layer-5 statistics pass
# also writes the layer-2 activation as a bf16 [V, 32] row (+64 bytes per view written once); the layer-6 statistics pass,
# the score pass and stage 6 of the backward start from that row (64 instead of 32 bytes per view read, layers 1 and 2 not
# evaluated).  Same numbers bit for bit: the row is the operand layer 5 consumes.  A/B: profiles/r06_hybrid_ab.json.
CHAIN_A2 = os.environ.get("DVA_CHAIN_A2", "0") == "1"


def chain_prologue(module, x_map, csr_idx, store_a2=False):
    """Everything of a chain forward that does not depend on the values: tile table, view -> point index, weight
    operands, the statistics passes of the four BatchNorm layers of DeepSetFeat (train mode), the set branch.
    Returns a namespace with the tensors the fused view kernel and the backward need (shared by the nearest path
    below and the bilinear path of fused_bilinear.py)."""
    from types import SimpleNamespace
    lib = _lib.load()
    e_map, e_score, gate = module.E_map, module.E_score, module.G
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
    gw = gate.weight.detach().reshape(-1).float().contiguous() if gate is not None else None
    gb = gate.bias.detach().reshape(-1).float().contiguous() if gate is not None else None

    zpool = iter(ops.zeros_small((11, 3 * D), torch.float64, dev))   # sum | sum of squares | input sums (+ moments)

    def zstats():
        return next(zpool)

    with ops._timed("chain_tiles", N * 8):
        tiles, n_tiles = build_tiles(csr_idx, V)
        vp = torch.empty(V, dtype=torch.int32, device=dev)
        check(lib.dva_csr_expand(ptr(csr_idx), N, ptr(vp), st), "dva_csr_expand")
    wops = torch.empty(OPS_BYTES, dtype=torch.uint8, device=dev)
    check(lib.dva_chain_prep(ptr(W1), ptr(W2), ptr(W5), W5.shape[1], ptr(W6), ptr(Ws), G, ptr(wops), st),
          "dva_chain_prep")
    # ---- layer 1: statistics from the moments of x_map
    s1 = zstats()
    mom = zstats()[:44]
    if training:
        with ops._timed("chain_moments", V * 32):
            check(lib.dva_chain_moments(ptr(x_map), V, ptr(W1), 0, ptr(mom), ptr(s1), st), "dva_chain_moments")
    bn1 = _chain_bn(s1, V, bns[0], training, W1, 8, mom)           # mom[:8] = sum of x_map
    # ---- layer 2: statistics + set pooling
    s2 = zstats()
    zstar = torch.empty((N, D), dtype=torch.float32, device=dev)
    arg = torch.empty((N, D), dtype=torch.int32, device=dev)
    with ops._timed("chain_stats2", V * 36 + N * 256):
        check(lib.dva_chain_stats2(ptr(x_map), ptr(vp), ptr(tiles), ptr(n_tiles), ptr(wops), ptr(bn1),
                                   ptr(bns[1].weight.detach()), ptr(s2), ptr(zstar), ptr(arg), V, st),
              "dva_chain_stats2")
    bn2 = _chain_bn(s2, V, bns[1], training, W2, D, s2[2 * D:])    # stats2 also sums the layer's input a1
    pooled = torch.empty((N, D), dtype=torch.float32, device=dev)
    check(lib.dva_chain_pooled(ptr(zstar), ptr(bn2), ptr(csr_idx), ptr(pooled), N, st), "dva_chain_pooled")
    t_add, set_saved = _set_branch_forward(e_map, pooled, csr_idx, training, zstats)
    # ---- layers 5, 6: statistics (train mode)
    s5, s6 = zstats(), zstats()
    a2 = None
    if training and store_a2 and V * 64 <= 0xfffffff0:
        a2 = torch.empty((V, D), dtype=torch.bfloat16, device=dev)
    if training:
        with ops._timed("chain_stats5", V * (36 + (64 if a2 is not None else 0)) + N * 128):
            if a2 is not None:
                check(lib.dva_chain_stats_a2(5, ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                             ptr(bn1), ptr(bn2), None, ptr(s5), V, N, ptr(a2), st), "dva_chain_stats_a2")
            else:
                check(lib.dva_chain_stats(5, ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                          ptr(bn1), ptr(bn2), None, ptr(s5), V, N, st), "dva_chain_stats")
    bn5 = _chain_bn(s5, V, bns[2], training)                       # layer 5 is not folded
    if training:
        with ops._timed("chain_stats6", V * (68 if a2 is not None else 36) + N * 128):
            if a2 is not None:
                check(lib.dva_chain_stats_a2(6, None, ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                             ptr(bn1), ptr(bn2), ptr(bn5), ptr(s6), V, N, ptr(a2), st), "dva_chain_stats_a2")
            else:
                check(lib.dva_chain_stats(6, ptr(x_map), ptr(vp), ptr(t_add), ptr(tiles), ptr(n_tiles), ptr(wops),
                                          ptr(bn1), ptr(bn2), ptr(bn5), ptr(s6), V, N, st), "dva_chain_stats")
    bn6 = _chain_bn(s6, V, bns[3], training, W6, D, s6[2 * D:])
    return SimpleNamespace(vp=vp, tiles=tiles, n_tiles=n_tiles, wops=wops, t_add=t_add, zstar=zstar, arg=arg, mom=mom,
                           bn1=bn1, bn2=bn2, bn5=bn5, bn6=bn6, bs=bs, gw=gw, gb=gb, W1=W1, G=G, training=training,
                           set_saved=set_saved, a2=a2)


# order of the chain's tensors in ctx.saved_tensors (after the path's own): tests/test_gpu_chain.py reads bn1 ..., scores
CHAIN_SAVED = ("vp", "tiles", "n_tiles", "wops", "t_add", "zstar", "arg", "mom", "bn1", "bn2", "bn5", "bn6")


class _ChainPool(torch.autograd.Function):
    """params: Wa, g1, b1, Wb, g2, b2, Wc, g3, b3, Wd, g4, b4, Ws, bs, gate_w, gate_b (or None), mlp_set params."""

    @staticmethod
    def forward(ctx, rows, row_idx, plan, x_map, csr_idx, module, scaling, eps, *params):
        lib = _lib.load()
        require_device(rows, row_idx, x_map, csr_idx)
        rows = rows.contiguous()
        x_map = x_map.contiguous()
        dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
        R, C = rows.shape
        st = stream_of(x_map)
        need_bwd = any(ctx.needs_input_grad)
        S = chain_prologue(module, x_map, csr_idx, store_a2=CHAIN_A2 and need_bwd)
        # ---- the fused view kernel
        out = pooled_output(csr_idx, N, C, dev)
        # a backward will follow: the scores of every view stay (16 bytes per view) -- the attention backward starts from
        # them instead of evaluating the chain once more
        scores = torch.empty((V, 4), dtype=torch.float32, device=dev) if need_bwd else None
        # SURVEY.md 8(d) fused view-gather + attention: V (C s + F_map 4 + idx) + N (C s + ptr); idx = view->point
        # index + row index (4 + 4), per point the set-branch row (128) on top (+ 16 bytes per view of scores out in training)
        with ops._timed("chain_attn_fwd", V * (C * 2 + 32 + 8 + (16 if need_bwd else 0)) + N * (C * 2 + 128 + 8)):
            check(lib.dva_chain_attn_fwd(ptr(x_map), ptr(S.vp), ptr(S.t_add), ptr(S.tiles), ptr(S.n_tiles), ptr(S.wops),
                                         ptr(S.bn1), ptr(S.bn2), ptr(S.bn5), ptr(S.bn6), ptr(S.bs), ptr(rows),
                                         ptr(row_idx), ptr(csr_idx), ptr(S.gw), ptr(S.gb), ptr(out), ptr(scores), N, V,
                                         R, C, S.G, int(scaling), float(eps), st), "dva_chain_attn_fwd")
        ctx.save_for_backward(rows, row_idx, x_map, csr_idx, S.vp, S.tiles, S.n_tiles, S.wops, S.t_add, S.zstar, S.arg,
                              S.mom, S.bn1, S.bn2, S.bn5, S.bn6, out, scores, S.bs, S.gw, S.gb, S.W1)
        ctx.plan = plan
        ctx.module = module
        ctx.set_saved = S.set_saved
        ctx.a2 = S.a2          # a workspace of this step (like set_saved): released by the backward
        ctx.training = S.training
        ctx.meta = (int(scaling), float(eps))
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import fused_chain_bwd
        return fused_chain_bwd.backward(ctx, gout)


def chain_params(module):
    """The parameters of the chain in the order of the gradients fused_chain_bwd.chain_epilogue returns."""
    e_map, e_score, gate = module.E_map, module.E_score, module.G
    blocks = (e_map.mlp_elt_1[0], e_map.mlp_elt_1[1], e_map.mlp_elt_2[0], e_map.mlp_elt_2[1])
    params = []
    for blk in blocks:
        bn = _bn_of(blk)
        params += [blk[0].weight, bn.weight, bn.bias]
    params += [e_score.weight, e_score.bias]
    params += [gate.weight, gate.bias] if gate is not None else [None, None]
    params += list(e_map.mlp_set.parameters())
    return params


def chain_pool(module, x_mod, x_map, csr_idx):
    """``module`` = GroupBimodalCSRPool, ``x_mod`` = ops.GatheredFeatures whose rows are already E_mod(rows)."""
    csr_idx = ops._check_ptr(csr_idx)
    return _ChainPool.apply(x_mod.rows, x_mod.row_idx.contiguous(), x_mod.plan, x_map, csr_idx, module,
                            module.group_scaling, 1e-12, *chain_params(module))


# ---------------------------------------------------------------------------------------------------------------------
# QKVBimodalCSRPool on the chain (round 4): the keys K(E_map(x_map)) are one more 32 x 32 layer behind DeepSetFeat.  Default
# (qkv_pool): the whole forward in the chain's view kernel (dva_chain_attn_fwd_keys); the bf16 key rows [V, 32] (accumulator
# order) and the compatibilities [V, 4] stay for the backward, where dQ comes from the key rows (dva_qkv_dquery) and the key
# gradient is built in registers inside the chain passes (dva_chain_score_stats_keys / dva_chain_bwd_layer6_keys).
# qkv_compatibilities: keys + compatibilities in one pass, attention on the scores-in kernels (DVA_QKV_ONE_KERNEL=0, A/B).
# ---------------------------------------------------------------------------------------------------------------------
_KEY_POS = {}


def key_position_order(device):
    """kappa [32]: key channel held by position i of a key row (i = 16 h + r -> (r & 3) + 8 (r >> 2) + 4 h)."""
    k = str(device)
    if k not in _KEY_POS:
        i = torch.arange(D)
        r, h = i % 16, i // 16
        _KEY_POS[k] = ((r & 3) + 8 * (r >> 2) + 4 * h).to(device)
    return _KEY_POS[k]


def keys_applicable(module, x_mod, x_map, csr_idx):
    """Can ``module`` (a QKVBimodalCSRPool) take its keys from the recompute chain?"""
    if not enabled() or module.use_mod_k or module.use_mod_q or module.save_last or module.debug:
        return False
    if not isinstance(x_mod, ops.GatheredFeatures) or x_mod.rows.dtype != torch.bfloat16:
        return False
    if module.K.out_features != D or module.num_groups not in (1, 2, 4) or module.nc_qk * module.num_groups != D:
        return False
    if not fused_deepset.applicable(module.E_map, module.K, x_map):
        return False
    V, N = x_map.shape[0], csr_idx.shape[0] - 1
    return V * 64 < (1 << 32) - 16 and N * 128 < (1 << 32) - 16


def keys_rows_ok(module, x_mod):
    """Second half of ``keys_applicable``, on the rows E_mod RETURNED: bf16, a width the view kernel is instantiated for
    per group, fp32 gate parameters (the kernels read them as float)."""
    rows = x_mod.rows
    C, G = rows.shape[1], module.num_groups
    if rows.dtype != torch.bfloat16 or C % G or (C // G) % 8:
        return False
    if module.G is not None and any(t is not None and t.dtype != torch.float32 for t in (module.G.weight, module.G.bias)):
        return False
    return rows.shape[0] * C * 2 < (1 << 32) - 16


class _KeyAdapter:
    """What chain_prologue / chain_epilogue read of a pooling module, for the key layer of a QKVBimodalCSRPool."""

    def __init__(self, module, with_gate=False):
        self.E_map, self.E_score, self.G = module.E_map, module.K, (module.G if with_gate else None)


class _ChainQKVPool(torch.autograd.Function):
    """QKVBimodalCSRPool.forward (pooling.py:500-547) in ONE view kernel: DeepSetFeat + key layer on the chain, compatibilities
    with the point's query row, softmax, gathered value rows, weighted sum, gate (dva_chain_attn_fwd_keys).  ``module`` = a
    _KeyAdapter with the gate; ``Qp`` fp32 [N, 32] = the queries in position order; params as chain_params(adapter)."""

    @staticmethod
    def forward(ctx, rows, row_idx, plan, x_map, csr_idx, module, scaling, eps, Qp, qk, *params):
        lib = _lib.load()
        require_device(rows, row_idx, x_map, csr_idx, Qp)
        rows, x_map, Qp = rows.contiguous(), x_map.contiguous(), Qp.float().contiguous()
        dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
        R, C = rows.shape
        groups, scale = qk
        st = stream_of(x_map)
        S = chain_prologue(module, x_map, csr_idx)
        out = pooled_output(csr_idx, N, C, dev)
        need_bwd = any(ctx.needs_input_grad)
        scores = torch.empty((V, 4), dtype=torch.float32, device=dev) if need_bwd else None
        keys = torch.empty((V, D), dtype=torch.bfloat16, device=dev) if need_bwd else None
        with ops._timed("chain_attn_fwd", V * (C * 2 + 32 + 8 + (16 + 64 if need_bwd else 0)) + N * (C * 2 + 256 + 8)):
            check(lib.dva_chain_attn_fwd_keys(ptr(x_map), ptr(S.vp), ptr(S.t_add), ptr(S.tiles), ptr(S.n_tiles), ptr(S.wops),
                                              ptr(S.bn1), ptr(S.bn2), ptr(S.bn5), ptr(S.bn6), ptr(S.bs), ptr(Qp), float(scale),
                                              ptr(rows), ptr(row_idx), ptr(csr_idx), ptr(S.gw), ptr(S.gb), ptr(out),
                                              ptr(scores), ptr(keys), N, V, R, C, int(groups), int(scaling), float(eps), st),
                  "dva_chain_attn_fwd_keys")
        ctx.save_for_backward(rows, row_idx, x_map, csr_idx, S.vp, S.tiles, S.n_tiles, S.wops, S.t_add, S.zstar, S.arg,
                              S.mom, S.bn1, S.bn2, S.bn5, S.bn6, out, scores, S.bs, S.gw, S.gb, S.W1, keys, Qp)
        ctx.plan, ctx.module, ctx.set_saved, ctx.training = plan, module, S.set_saved, S.training
        ctx.meta = (int(scaling), float(eps))
        ctx.qk = (int(groups), float(scale))
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import fused_chain_bwd
        return fused_chain_bwd.backward(ctx, gout)


class _ChainCompat(torch.autograd.Function):
    """compatibilities fp32 [V, 4] (the first ``groups`` columns used) = scale * group sums of K(E_map(x_map)) * Q'[point]:
    the key layer as one more layer of the recompute chain, the products with the point's query row in the same kernel
    (dva_chain_keys_compat); the bf16 key rows [V, 32] (position order) stay for dQ'.  ``Qp`` fp32 [N, 32] = the queries in
    position order; params in fused_chain.chain_params(adapter) order."""

    @staticmethod
    def forward(ctx, x_map, csr_idx, Qp, adapter, groups, scale, *params):
        lib = _lib.load()
        require_device(x_map, csr_idx, Qp)
        x_map = x_map.contiguous()
        Qp = Qp.float().contiguous()
        dev, V, N = x_map.device, x_map.shape[0], csr_idx.shape[0] - 1
        st = stream_of(x_map)
        S = chain_prologue(adapter, x_map, csr_idx)
        keys = torch.empty((V, D), dtype=torch.bfloat16, device=dev)
        compat = torch.empty((V, 4), dtype=torch.float32, device=dev)
        with ops._timed("chain_keys", V * (32 + 4 + 64 + 16) + N * 256):
            check(lib.dva_chain_keys_compat(ptr(x_map), ptr(S.vp), ptr(S.t_add), ptr(S.tiles), ptr(S.n_tiles), ptr(S.wops),
                                            ptr(S.bn1), ptr(S.bn2), ptr(S.bn5), ptr(S.bn6), ptr(S.bs), ptr(keys), ptr(Qp),
                                            ptr(compat), int(groups), float(scale), V, N, st), "dva_chain_keys_compat")
        ctx.save_for_backward(x_map, csr_idx, S.vp, S.tiles, S.n_tiles, S.wops, S.t_add, S.zstar, S.arg, S.mom, S.bn1,
                              S.bn2, S.bn5, S.bn6, S.W1, keys, Qp)
        ctx.adapter, ctx.set_saved, ctx.training = adapter, S.set_saved, S.training
        ctx.meta = (int(groups), float(scale))
        return compat

    @staticmethod
    def backward(ctx, dcompat):
        from types import SimpleNamespace
        from .fused_chain_bwd import Arena, chain_epilogue
        lib = _lib.load()
        if ctx.set_saved is None:
            raise RuntimeError("the recompute chain's backward ran twice on the same graph (retain_graph is not "
                               "supported on this path)")
        (x_map, csr_idx, vp, tiles, n_tiles, wops, t_add, zstar, arg, mom, bn1, bn2, bn5, bn6, W1, keys,
         Qp) = ctx.saved_tensors
        groups, scale = ctx.meta
        V, N = x_map.shape[0], csr_idx.shape[0] - 1
        dcompat = dcompat.float().contiguous()
        dQ = torch.empty((N, D), dtype=torch.float32, device=x_map.device)
        with ops._timed("qkv_dquery", V * (64 + 16) + N * 136):
            check(lib.dva_qkv_dquery(ptr(dcompat), 4, ptr(keys), ptr(csr_idx), ptr(dQ), N, V, groups, scale,
                                     stream_of(x_map)), "dva_qkv_dquery")
        S = SimpleNamespace(vp=vp, tiles=tiles, n_tiles=n_tiles, wops=wops, t_add=t_add, zstar=zstar, arg=arg, mom=mom,
                            bn1=bn1, bn2=bn2, bn5=bn5, bn6=bn6, W1=W1, G=D, training=ctx.training)
        grads = chain_epilogue(lib, Arena(x_map.device), S, ctx.adapter, x_map, csr_idx, dcompat, None, ctx.set_saved,
                               keys=(Qp, groups, scale))
        ctx.set_saved = None
        return (None, None, dQ if ctx.needs_input_grad[2] else None, None, None, None) + tuple(grads)


class _QKCompat(torch.autograd.Function):
    """compat [V, G] from key rows (position order) and per-point queries (position order): the stand-alone kernels of
    csrc/qkv.hip (the chain path computes the compatibilities inside dva_chain_keys_compat; kept for A/B and tests)."""

    @staticmethod
    def forward(ctx, keys, Qp, csr_idx, vp, G, scale):
        lib = _lib.load()
        V = keys.shape[0]
        Qp = Qp.float().contiguous()
        compat = torch.empty((V, G), dtype=torch.float32, device=keys.device)
        with ops._timed("qkv_compat", V * (64 + 128 + 4 * G)):
            check(lib.dva_qkv_compat(ptr(keys), ptr(Qp), ptr(vp), ptr(compat), V, G, float(scale), stream_of(keys)),
                  "dva_qkv_compat")
        ctx.save_for_backward(keys, Qp, csr_idx, vp)
        ctx.meta = (G, float(scale))
        return compat

    @staticmethod
    def backward(ctx, dcompat):
        lib = _lib.load()
        keys, Qp, csr_idx, vp = ctx.saved_tensors
        G, scale = ctx.meta
        V, N = keys.shape[0], Qp.shape[0]
        dcompat = dcompat.float().contiguous()
        dkeys = torch.empty((V, D), dtype=torch.bfloat16, device=keys.device)
        dQ = torch.empty((N, D), dtype=torch.float32, device=keys.device)
        with ops._timed("qkv_compat_bwd", V * (64 + 64 + 128 + 8 * G) + N * 128):
            check(lib.dva_qkv_compat_bwd(ptr(dcompat), ptr(keys), ptr(Qp), ptr(vp), ptr(csr_idx), ptr(dkeys), ptr(dQ), N, V,
                                         G, scale, stream_of(keys)), "dva_qkv_compat_bwd")
        return dkeys, dQ, None, None, None, None


def qkv_compatibilities(module, x_main, x_map, csr_idx):
    """``compatibilities`` of QKVBimodalCSRPool.forward (pooling.py:520-531) for point-wise queries and mapping-feature
    keys: ``x_main`` = E_main(x_main) [N, nc_inner]."""
    import math
    csr_idx = ops._check_ptr(csr_idx)
    adapter = _KeyAdapter(module)
    kappa = key_position_order(x_map.device)
    # the point's queries in the key rows' position order: the permutation goes onto the [32, 32] weight, not onto [N, 32]
    Qp = ops.tall_linear(x_main, module.Q.weight[kappa], module.Q.bias[kappa]).float()
    scale = 1.0 / math.sqrt(module.nc_qk) if module.dim_scaling else 1.0
    G = module.num_groups
    compat = _ChainCompat.apply(x_map, csr_idx, Qp, adapter, G, scale, *chain_params(adapter))
    return compat if G == 4 else compat[:, :G]


# DVA_QKV_ONE_KERNEL=0: keys + compatibilities in one pass, attention in the scores-in kernels (the A/B of round 4)
QKV_ONE_KERNEL = os.environ.get("DVA_QKV_ONE_KERNEL", "1") == "1"


def qkv_pool(module, x_main, x_mod, x_map, csr_idx):
    """The whole QKVBimodalCSRPool.forward behind E_main / E_mod for point-wise queries and mapping-feature keys:
    ``x_main`` = E_main(x_main) [N, nc_inner], ``x_mod`` = ops.GatheredFeatures whose rows are already E_mod(rows)."""
    import math
    csr_idx = ops._check_ptr(csr_idx)
    adapter = _KeyAdapter(module, with_gate=True)
    kappa = key_position_order(x_map.device)
    Qp = ops.tall_linear(x_main, module.Q.weight[kappa], module.Q.bias[kappa]).float()
    scale = 1.0 / math.sqrt(module.nc_qk) if module.dim_scaling else 1.0
    return _ChainQKVPool.apply(x_mod.rows, x_mod.row_idx.contiguous(), x_mod.plan, x_map, csr_idx, adapter,
                               module.group_scaling, 1e-12, Qp, (module.num_groups, scale), *chain_params(adapter))
