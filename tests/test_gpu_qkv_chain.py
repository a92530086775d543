"""-m gpu: QKVBimodalCSRPool on the bf16 recompute chain (round 4: dva_chain_attn_fwd_keys = the whole forward in one view
kernel, dva_chain_keys_compat + the scores-in attention as its A/B, dva_qkv_dquery, the key-layer entries
dva_chain_score_stats_keys / dva_chain_bwd_layer6_keys) against the CPU oracle (reference modules/multimodal/pooling.py:
454-547).  Gates as for the group pooling (tests/test_gpu_chain.py): output <= max(2e-2, 1.5 x the oracle's own error under
torch.autocast(bfloat16)); gradients <= max(2 x autocast (4 x for gate parameters), 5e-2) per tensor, the yardstick of the
encoder parameters floored by its median over them; every gradient must exist."""
import pytest
import torch

from oracle import pooling_oracle as O
from test_gpu_chain import rel, make_case, ragged, ragged_long, full32

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(case, G, nc_qk, train, in_main=6, seed=7, wscale=0.3, **kw):
    from deepviewagg_amd.modules.multimodal import pooling as P
    gen = torch.Generator().manual_seed(seed)
    kwargs = dict(in_main=in_main, in_map=8, in_mod=case["C"], num_groups=G, nc_qk=nc_qk, use_num=True, **kw)
    ref = O.QKVBimodalCSRPool(**kwargs)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * wscale)
            if "batch_norm.weight" in n or n == "G.weight":
                p.add_(1.0)
        for n, b in ref.named_buffers():
            if "running_mean" in n:
                b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
            if "running_var" in n:
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)
    ref.train(train)
    m = P.QKVBimodalCSRPool(**kwargs)
    m.load_state_dict(ref.state_dict(), strict=True)
    return ref, m.to(DEV).train(train)


def run_dev(case, m, x_main, chain, one_kernel=True):
    from deepviewagg_amd import ops, fused_chain
    from deepviewagg_amd.modules.multimodal import pooling as P
    V = case["V"]
    xd = case["x"].to(DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    xm = x_main.to(DEV).requires_grad_()
    packed = ops.pack_gather_index(case["images"].to(DEV), torch.arange(V + 1, device=DEV), case["pixels"].to(DEV))
    fused_chain.FORCE = None if chain else False
    calls = []
    orig, orig_pool, orig_flag = fused_chain.qkv_compatibilities, fused_chain.qkv_pool, fused_chain.QKV_ONE_KERNEL

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)

    def spy_pool(*a, **k):
        calls.append(1)
        return orig_pool(*a, **k)
    fused_chain.qkv_compatibilities, fused_chain.qkv_pool, fused_chain.QKV_ONE_KERNEL = spy, spy_pool, one_kernel
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lazy = ops.lazy_gather_nearest(xd, packed, exact=True)
            lazy = P.BimodalCSRPool(mode='max')(None, lazy, None, torch.arange(V + 1, device=DEV))
            out = m(xm, lazy, case["x_map"].to(DEV), case["csr"].to(DEV))
        grads = torch.autograd.grad((out.float() * case["w"].to(DEV)).sum(), [xd, xm] + list(m.parameters()),
                                    allow_unused=True)
    finally:
        fused_chain.FORCE = None
        fused_chain.qkv_compatibilities, fused_chain.qkv_pool, fused_chain.QKV_ONE_KERNEL = orig, orig_pool, orig_flag
    return out, grads, len(calls)


def oracle(case, ref, x_main, autocast):
    xr = case["x"].clone().requires_grad_()
    xm = x_main.clone().requires_grad_()

    def fwd():
        return ref(xm, O.gather_nearest(xr, case["images"], case["pixels"]), case["x_map"], case["csr"])
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = fwd()
    else:
        out = fwd()
    grads = torch.autograd.grad((out.float() * case["w"]).sum(), [xr, xm] + list(ref.parameters()), allow_unused=True)
    return out, grads


@pytest.mark.parametrize("sizes_fn,N,C,G,nc_qk,train", [
    (ragged, 3000, 64, 4, 8, True),
    (ragged_long, 2000, 64, 4, 8, True),         # points with 33 / 64 / 70 / 100 views: fragment tiles
    (full32, 1024, 64, 4, 8, True),              # the S1 shape
    (ragged, 2000, 128, 2, 16, True),
    (ragged, 3000, 64, 1, 32, False),
    (ragged_long, 1500, 32, 2, 16, False),
    (ragged_long, 3000, 64, 1, 32, True),
    (ragged, 2500, 512, 4, 8, False),
])
def test_qkv_pool_on_the_chain_matches_oracle(sizes_fn, N, C, G, nc_qk, train):
    case = make_case(17, N, C, sizes_fn)
    gen = case["gen"]
    x_main = torch.randn(N, 6, generator=gen)
    ref, m = build(case, G, nc_qk, train)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    out_ref, g_ref = oracle(case, ref, x_main, autocast=False)
    ref.load_state_dict(sd)
    out_amp, g_amp = oracle(case, ref, x_main, autocast=True)
    out, g, n_chain = run_dev(case, m, x_main, chain=True)
    assert n_chain == 1, "the key layer must have run on the recompute chain"
    assert out.dtype == torch.bfloat16
    r, r_amp = rel(out, out_ref), rel(out_amp, out_ref)
    print(f"qkv chain fwd rel err {r:.4f} (oracle under autocast {r_amp:.4f})")
    assert r < max(2e-2, 1.5 * r_amp), (r, r_amp)
    unseen = case["csr"][1:] == case["csr"][:-1]
    assert float(out.detach().float().cpu()[unseen].abs().max() if unseen.any() else 0.0) == 0.0
    if train:
        for (k, a), b in zip(m.state_dict().items(), ref.state_dict().values()):
            if "running" in k:
                torch.testing.assert_close(a.cpu(), b, rtol=2e-2, atol=2e-3)
    names = ["x", "x_main"] + [n for n, _ in ref.named_parameters()]
    enc = ("E_map", "E_main", "Q.", "K.")
    amps = sorted(rel(c, b) for n, b, c in zip(names, g_ref, g_amp) if b is not None and n.startswith(enc))
    med = amps[len(amps) // 2]
    report, bad = [], []
    for n, a, b, c in zip(names, g, g_ref, g_amp):
        if b is None:
            assert a is None or float(a.abs().max()) == 0, n
            continue
        assert a is not None, f"no gradient for {n}"
        ours, amp = rel(a, b), rel(c, b)
        report.append((n, round(ours, 4), round(amp, 4)))
        if n.startswith(enc):
            amp = max(amp, med)
        if ours > max((4.0 if n.startswith("G.") else 2.0) * amp, 5e-2):
            bad.append(report[-1])
    print("qkv chain bwd rel err (ours, oracle under autocast):", report)
    assert not bad, (bad, report)
    # A/B against the stored-activation keys (the path of rounds 1-3) on the same inputs
    m.load_state_dict(sd)
    out_b, g_b, n_b = run_dev(case, m, x_main, chain=False)
    assert n_b == 0
    assert rel(out, out_b) < 2e-2, rel(out, out_b)
    assert rel(g[0], g_b[0]) < 1e-1, rel(g[0], g_b[0])
    # A/B of the two chain forms: everything in ONE view kernel (the default above) against keys + compatibilities in one
    # pass and the scores-in attention kernels -- the same arithmetic up to summation order
    m.load_state_dict(sd)
    out_c, g_c, n_c = run_dev(case, m, x_main, chain=True, one_kernel=False)
    assert n_c == 1
    assert rel(out, out_c) < 4e-3, rel(out, out_c)
    assert rel(g[0], g_c[0]) < 2e-2, rel(g[0], g_c[0])
    for n_, a, c_ in zip(names[2:], g[2:], g_c[2:]):
        if a is not None and c_ is not None and float(c_.float().norm()) > 0:
            assert rel(a, c_) < (1.5e-1 if train else 3e-2), (n_, rel(a, c_))


def test_qkv_pool_without_gate_and_without_dim_scaling():
    """gating=False (no Gating module: NULL gate pointers through the view kernel and the attention backward) and
    dim_scaling=False (scale 1) on ragged points: forward against the oracle, every existing gradient finite and close."""
    case = make_case(23, 2500, 64, ragged)
    x_main = torch.randn(2500, 6, generator=case["gen"])
    ref, m = build(case, 4, 8, True, gating=False, dim_scaling=False)
    assert m.G is None
    out_ref, g_ref = oracle(case, ref, x_main, autocast=False)
    out, g, n_chain = run_dev(case, m, x_main, chain=True)
    assert n_chain == 1
    assert rel(out, out_ref) < 2e-2, rel(out, out_ref)
    assert rel(g[0], g_ref[0]) < 6e-2, rel(g[0], g_ref[0])                 # feature-map gradient
    names = ["x", "x_main"] + [n for n, _ in ref.named_parameters()]
    for n, a, b in zip(names, g, g_ref):
        if b is None:
            continue
        assert a is not None and bool(torch.isfinite(a.float()).all()), n
        # (K.bias: without a gate the softmax is invariant to a per-point shift of the scores, its true gradient is 0 --
        #  the oracle's is 1e-6 of rounding noise, ours the bf16 noise of the same cancellation)
        if n.startswith(("E_mod", "Q.", "K.weight")):
            assert rel(a, b) < 2.5e-1, (n, rel(a, b))
        if n == "K.bias":
            assert float(a.float().norm()) < 0.05 * float(g[names.index("K.weight")].float().norm()), n


def test_qkv_compat_kernels_against_torch():
    """dva_qkv_compat / _bwd on random key rows: the position-order bookkeeping (group of a position, query permutation)
    against the module's own expression on channel-order tensors."""
    import math
    from deepviewagg_amd import fused_chain, ops
    gen = torch.Generator().manual_seed(3)
    for G in (4, 2, 1):
        nc = 32 // G
        sizes = ragged_long(700, gen)
        csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(DEV)
        V, N = int(csr[-1]), 700
        keys_c = torch.randn(V, 32, generator=gen).bfloat16().to(DEV)                 # channel order
        Q = torch.randn(N, 32, generator=gen).to(DEV).requires_grad_()
        kappa = fused_chain.key_position_order(DEV)
        keys_p = keys_c[:, kappa].contiguous().requires_grad_()
        vp = ops.csr_expand(csr, V)
        scale = 1 / math.sqrt(nc)
        compat = fused_chain._QKCompat.apply(keys_p, Q[:, kappa], csr, vp, G, scale)
        kf = keys_c.float().requires_grad_()
        Qr = Q.detach().clone().requires_grad_()
        ref = (kf.view(V, G, nc) * Qr[vp.long()].view(V, G, nc)).sum(2) * scale
        torch.testing.assert_close(compat, ref, rtol=1e-5, atol=1e-5)
        w = torch.randn(V, G, generator=gen).to(DEV)
        dk, dq = torch.autograd.grad((compat * w).sum(), [keys_p, Q])
        dk_r, dq_r = torch.autograd.grad((ref * w).sum(), [kf, Qr])
        inv = torch.argsort(kappa)
        assert rel(dk.float()[:, inv], dk_r) < 6e-3            # bf16 rows
        torch.testing.assert_close(dq, dq_r, rtol=1e-4, atol=1e-4)
        # dQ' alone from the padded [V, 4] gradient layout of the chain path (dva_qkv_dquery, ld = 4)
        from deepviewagg_amd import _lib
        from deepviewagg_amd._lib import check, ptr, stream_of
        w4 = torch.zeros((V, 4), device=DEV)
        w4[:, :G] = w
        dq4 = torch.empty((N, 32), device=DEV)
        check(_lib.load().dva_qkv_dquery(ptr(w4), 4, ptr(keys_p.detach()), ptr(csr), ptr(dq4), N, V, G, scale,
                                         stream_of(w4)), "dva_qkv_dquery")
        torch.testing.assert_close(dq4[:, inv], dq_r, rtol=1e-4, atol=1e-4)
