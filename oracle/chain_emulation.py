"""bf16-emulation oracle of the recompute chain (TEST INFRASTRUCTURE ONLY: imported by tests/).

GroupBimodalCSRPool.forward of the reference (modules/multimodal/pooling.py:263-315 with DeepSetFeat :658-669, MLP block
core/common_modules/base_modules.py:38-48) with the operand roundings of csrc/chain_*.hip made explicit, so that a
comparison with the kernels is not separated by anything but fp32 summation order."""
import torch
import torch.nn.functional as F

from . import pooling_oracle as O


def _bf(t):
    return t + (t.bfloat16().float() - t).detach()


def emulated_chain(ref, vals, x_map, csr, dev_invstd=None, dev_scores=None, return_scores=False):
    """``dev_invstd``: optional {layer: fp32 [32]} = the BatchNorm invstd of the folded layers (1, 2, 6) AS THE DEVICE
    COMPUTED IT.  The folded operand bf16(0.6 gamma invstd W) is a discontinuous function of the batch statistics: an
    invstd that differs in its last bit (another summation order) flips the rounding of an entry here and there, and
    one flipped entry (2^-8 of one weight, seen by every view) moves train-mode parameter gradients by ~10 %
    (tests/test_oracle_chaos.py).  With ``dev_invstd`` the ROUNDING DECISIONS of the operand are taken from the
    device's constants (same fp32 expression as csrc/chain_common.h fold_ops), while value and gradient still flow
    through the emulation's own statistics: both sides then evaluate the same discrete network.

    ``dev_scores``: optional fp32 [V, G] = the scores E_score(E_map(x_map)) as the device computed them (they differ
    from the emulation's by ~1e-3 relative: fp32 summation order through five layers).  The attention tail has two
    kinks that turn such a difference into O(1) changes of single gradient entries: the arg-max view of a point takes
    the whole gate gradient, and the gate tanh(relu(w m + b)) switches its derivative on at 0 -- two or three points
    of 1500 straddle it, and their score gradients are an order of magnitude larger than the typical entry (measured:
    8 % of the L2 norm of the score gradient from 2 points, tools/debug_chain.py).  With ``dev_scores`` the tail is
    evaluated AT the device's scores (straight-through: value from the device, gradient into the emulation's own
    scores), so both sides take the same branches.

    GroupBimodalCSRPool.forward of the oracle (pooling.py:263-315, :658-669) given the per-view values
    ``vals`` = E_mod(x_mod) [V, C], with the chain's roundings.  A layer whose raw output a pass does not need is
    evaluated with BatchNorm folded into the rounded weight operand: t = a . bf16(0.6 G W)^T + 0.6 B,
    leaky(y) = t + (2/3) |t|; G comes from the statistics of the plain product a . bf16(W)^T (what the statistics
    pass of that layer sees; the set pooling takes its max there too), the shift from the batch mean of the folded
    product itself.  Layers 1, 2, 6 fold; layer 5 adds the
    per-point row before its BatchNorm and stays as it is."""
    E = ref.E_map
    idx = O.dense_index(csr)

    def bn_act(blk, z):
        return F.leaky_relu(blk[1](z), 0.2)

    def folded(blk, a_prev, layer):
        """(activation of the folded product, activation of the plain product)"""
        W, bn = blk[0].weight, blk[1].batch_norm
        z = a_prev @ _bf(W).t()
        if bn.training:
            mean, var = z.mean(0), z.var(0, unbiased=False)
        else:
            mean, var = bn.running_mean, bn.running_var
        g = bn.weight * torch.rsqrt(var + bn.eps)
        if dev_invstd is not None and layer in dev_invstd:
            s = (torch.tensor(0.6, dtype=torch.float32) * bn.weight.detach()) * dev_invstd[layer]   # fold_ops: 0.6f * gamma * invstd
            exact = 0.6 * g.view(-1, 1) * W
            Wf = exact + ((W.detach() * s.view(-1, 1)).bfloat16().float() - exact).detach()
        else:
            Wf = _bf(0.6 * g.view(-1, 1) * W)
        if bn.training:     # the shift keeps the exact batch mean of the folded product (dva_chain_bn_consts)
            shift = 0.6 * bn.bias - Wf @ a_prev.mean(0)
        else:
            shift = 0.6 * (bn.bias - mean * g)
        t = a_prev @ Wf.t() + shift
        return t + (2.0 / 3.0) * t.abs(), bn_act(blk, z)      # the module call updates the running statistics

    # the first layer sees x_map to 16 bits (hi | lo in the k-slots of one matrix-core instruction)
    x_hi = x_map.bfloat16().float()
    x16 = x_hi + (x_map - x_hi).bfloat16().float()
    a1 = _bf(folded(E.mlp_elt_1[0], x16, 1)[0])
    a2f, a2_plain = folded(E.mlp_elt_1[1], a1, 2)
    x_set = O.segment_csr(a2_plain, csr, 'max')
    if E.use_num:
        set_num = torch.sqrt(1 / (csr[1:] - csr[:-1] + 1e-3))
        x_set = torch.cat((x_set, set_num.view(-1, 1).float()), dim=1)
    s = E.mlp_set(x_set)
    Wc = E.mlp_elt_2[0][0].weight
    u = s @ Wc[:, 32:].t()
    a5 = _bf(bn_act(E.mlp_elt_2[0], _bf(a2f) @ _bf(Wc[:, :32]).t() + u[idx]))
    a6 = _bf(folded(E.mlp_elt_2[1], a5, 6)[0])
    compat = a6 @ _bf(ref.E_score.weight).t() + ref.E_score.bias
    own = compat
    if dev_scores is not None:
        compat = compat + (dev_scores - compat).detach()
    if return_scores and compat.requires_grad:
        compat.retain_grad()
    out, _, _ = O.attention_tail(vals, compat, csr, ref.G, ref.num_groups, ref.out_mod, ref.group_scaling)
    return (out, compat if dev_scores is None else own) if return_scores else out


def emulated_emod(ref, x, images, pixels, mapping_size, stored_za=None):
    """E_mod of the fused BILINEAR path with the roundings of csrc/chain_emod.hip made explicit (round 5; VERDICT r4
    item 7): the per-view values ``E_mod(sparse_interpolation(x))`` [V, C_o] that ``emulated_chain`` takes as ``vals``.

    Reference: sparse_interpolation (core/multimodal/image.py:105-170) -> E_mod = [Linear_a, BN_a, LeakyReLU, Linear_b,
    BN_b, LeakyReLU] per view (modules/multimodal/pooling.py:245, :275).  Device arithmetic (fused_bilinear.py):
      * Linear_a commutes with the interpolation: Y = x_rows . bf16(W_a)^T on the MAP rows (fp32 accumulation), stored as
        bf16 -- the rounding the layer's output has under autocast;
      * z_a[v] = sum_k w_k Y[tap_k(v)] in fp32 from the bf16 rows; TRAIN mode keeps it as bf16 [V, C_o] (every later pass
        reads the stored row) and BatchNorm_a's batch statistics are those of the STORED values; eval mode uses the fp32
        value (one kernel, nothing stored);
      * y_a = leaky(BN_a(z_a)) enters Linear_b as a bf16 operand against bf16(W_b), fp32 accumulation;
      * BatchNorm_b (not folded into the operand), LeakyReLU and the attention-weighted sum stay fp32.
    ``x`` fp32 [B, C_in, H, W] with values on the bf16 grid (the feature maps are bf16 on the device).
    ``stored_za``: evaluate from the bf16 z_a (default: train mode).  A BACKWARD behind an eval-mode forward builds the
    stored row first and differentiates the evaluation from it (fused_bilinear._EmodPool.backward), so its gradients are
    those of ``stored_za=True`` although the eval forward itself used the fp32 row -- 2-3 % apart on the feature-map and
    E_mod gradients (leaky' flips of the values the rounding moves across zero; measured, tests/test_gpu_bilinear.py)."""
    lin_a, bn_a = ref.E_mod[0][0], ref.E_mod[0][1]
    lin_b, bn_b = ref.E_mod[1][0], ref.E_mod[1][1]
    B, C_in, H, W = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C_in)
    Y = _bf(rows @ _bf(lin_a.weight).t())                                   # [R, C_o] map rows of Linear_a
    y_map = Y.reshape(B, H, W, -1).permute(0, 3, 1, 2)
    z_a = O.gather_bilinear(y_map, images, pixels, mapping_size)            # fp32 interpolation of the bf16 rows
    if bn_a.batch_norm.training if stored_za is None else stored_za:
        z_a = _bf(z_a)                                                      # the stored row; its statistics
    y_a = F.leaky_relu(bn_a(z_a), 0.2)
    z_b = _bf(y_a) @ _bf(lin_b.weight).t()
    return F.leaky_relu(bn_b(z_b), 0.2)
