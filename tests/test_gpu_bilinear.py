"""-m gpu: the fused bilinear path (interpolate=True; csrc/chain_emod.hip through fused_bilinear.py) against the CPU
oracle (sparse_interpolation + per-view E_mod + GroupBimodalCSRPool, reference core/multimodal/image.py:105-170,
modules/multimodal/pooling.py:263-315) and against the materialised device dataflow (gather_bilinear -> [V, C] ->
E_mod through the row kernels -> first-generation attention kernels).

Tolerances (bf16 operands on the matrix cores, like the reference under torch.autocast(bfloat16)): output relative
L2 <= max(2e-2, 1.5 x the oracle's own autocast error); gradients <= max(2 x (4 x gate / score) autocast error, 5e-2)
per tensor, the yardstick of the E_map / E_mod parameters floored by its median over them (tests/test_gpu_chain.py)."""
import pytest
import torch

from oracle import pooling_oracle as O
from test_gpu_chain import rel, ragged, ragged_long, full32

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
UP = 8          # mapping resolution = UP x feature-map resolution


def make_case(seed, N, C_in, sizes_fn, B=3, H=12, W=20):
    gen = torch.Generator().manual_seed(seed)
    sizes = sizes_fn(N, gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    images = torch.randint(0, B, (V,), generator=gen)
    msize = (W * UP, H * UP)
    pixels = torch.stack([torch.randint(0, msize[0], (V,), generator=gen),
                          torch.randint(0, msize[1], (V,), generator=gen)], 1).short()
    # image borders (replicated padding) are hit on purpose
    pixels[:8, 0] = torch.tensor([0, 0, msize[0] - 1, msize[0] - 1, 1, msize[0] - 2, 0, 5])
    pixels[:8, 1] = torch.tensor([0, msize[1] - 1, 0, msize[1] - 1, 1, msize[1] - 2, 7, 0])
    x = torch.randn(B, C_in, H, W, generator=gen).bfloat16().float()
    x_map = torch.rand(V, 8, generator=gen)
    return dict(gen=gen, csr=csr, V=V, images=images, pixels=pixels, x=x, x_map=x_map, N=N, C=C_in, msize=msize)


def build(case, C_out, G, train, gating=True, seed=5, wscale=0.3):
    from deepviewagg_amd.modules.multimodal import pooling as P
    gen = torch.Generator().manual_seed(seed)
    kwargs = dict(in_map=8, in_mod=case["C"], out_mod=C_out, num_groups=G, use_num=True, gating=gating)
    ref = O.GroupBimodalCSRPool(**kwargs)
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
    m = P.GroupBimodalCSRPool(**kwargs)
    m.load_state_dict(ref.state_dict(), strict=True)
    return ref, m.to(DEV).train(train)


def run_dev(case, m, w, fused, need_grad=True):
    from deepviewagg_amd import ops, fused_chain
    from deepviewagg_amd.modules.multimodal import pooling as P
    V = case["V"]
    xd = case["x"].to(DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(need_grad)
    atom_ptr = torch.arange(V + 1, device=DEV)
    packed = ops.pack_gather_index(case["images"].to(DEV), atom_ptr, case["pixels"].to(DEV))
    res = torch.tensor([case["msize"]], dtype=torch.float32, device=DEV)
    coords = (case["pixels"].to(DEV) / (res - 1))[:, [1, 0]]
    fused_chain.FORCE = None if fused else False
    used = {}
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lazy = ops.lazy_gather_bilinear(xd, packed, coords, exact=True)
            lazy = P.BimodalCSRPool(mode='max')(None, lazy, None, atom_ptr)
            assert isinstance(lazy, ops.InterpolatedFeatures), "the atomic pool of an exact mapping stays lazy"
            out = m(None, lazy, case["x_map"].to(DEV), case["csr"].to(DEV))
        used["fn"] = type(out.grad_fn).__name__ if out.grad_fn is not None else None
        grads = None
        if need_grad:
            grads = torch.autograd.grad((out.float() * w.to(DEV)).sum(), [xd] + list(m.parameters()), allow_unused=True)
    finally:
        fused_chain.FORCE = None
    return out, grads, used


def oracle(case, ref, w, autocast):
    xr = case["x"].clone().requires_grad_()

    def fwd():
        xm = O.gather_bilinear(xr, case["images"], case["pixels"], case["msize"])
        return ref(None, xm, case["x_map"], case["csr"])
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = fwd()
    else:
        out = fwd()
    grads = torch.autograd.grad((out.float() * w).sum(), [xr] + list(ref.parameters()), allow_unused=True)
    return out, grads


CASES = [
    (ragged, 3000, 64, 64, 4, True),
    (full32, 512, 64, 64, 4, True),            # the headline shape: 32 views per point
    (ragged_long, 1500, 128, 32, 4, True),     # the KITTI-360 pair (l0: 128 -> 32), points with > 32 views
    (ragged_long, 1200, 64, 32, 2, False),
    (ragged, 900, 48, 64, 1, True),
    (ragged_long, 1000, 64, 64, 2, True),
    # wide rows (round 4): C_o = 128 trains on the block-by-block kernels (KITTI-360 pyramid level 256 -> 128)
    (ragged_long, 1200, 256, 128, 4, True),
    (ragged, 900, 96, 128, 2, True),
    (full32, 512, 64, 128, 1, True),
    (ragged_long, 700, 128, 128, 4, False),
    # C_o = 256 (pyramid level 512 -> 256): dz_b handed over in HBM, dW_b by the block-cooperative kernel
    (ragged_long, 900, 160, 256, 4, True),
    (full32, 512, 64, 256, 4, True),
    (ragged_long, 1200, 96, 256, 4, False),
]


@pytest.mark.parametrize("sizes_fn,N,C_in,C_out,G,train", CASES)
def test_fused_bilinear_matches_oracle_and_materialised_path(sizes_fn, N, C_in, C_out, G, train):
    case = make_case(21, N, C_in, sizes_fn)
    w = torch.randn(N, C_out, generator=case["gen"])
    ref, m = build(case, C_out, G, train)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    out_ref, g_ref = oracle(case, ref, w, autocast=False)
    ref.load_state_dict(sd)
    out_amp, g_amp = oracle(case, ref, w, autocast=True)
    out, g, used = run_dev(case, m, w, fused=True)
    assert used["fn"] == "_EmodPoolBackward", f"the fused bilinear path must be the one that ran ({used})"
    assert out.dtype == torch.bfloat16
    r, r_amp = rel(out, out_ref), rel(out_amp, out_ref)
    print(f"fused bilinear fwd rel err {r:.4f} (oracle under autocast {r_amp:.4f})")
    assert r < max(2e-2, 1.5 * r_amp), (r, r_amp)
    unseen = case["csr"][1:] == case["csr"][:-1]
    assert float(out.detach().float().cpu()[unseen].abs().max() if unseen.any() else 0.0) == 0.0
    if train:
        for (k, a), b in zip(m.state_dict().items(), ref.state_dict().values()):
            if "running" in k:
                torch.testing.assert_close(a.cpu(), b, rtol=2e-2, atol=2e-3)
    # gradients against the fp32 oracle, yardstick = the oracle under autocast
    names = ["x"] + [n for n, _ in ref.named_parameters()]
    amps = sorted(rel(c, b) for n, b, c in zip(names, g_ref, g_amp)
                  if b is not None and (n.startswith("E_map") or n.startswith("E_mod")))
    med = amps[len(amps) // 2]
    report, bad = [], []
    for n, a, b, c in zip(names, g, g_ref, g_amp):
        if b is None:
            assert a is None or float(a.abs().max()) == 0, n
            continue
        assert a is not None, n
        ours, amp = rel(a, b), rel(c, b)
        report.append((n, round(ours, 4), round(amp, 4)))
        if n.startswith("E_map") or n.startswith("E_mod"):
            amp = max(amp, med)
        loose = n.startswith("G.") or n.startswith("E_score")
        if ours > max((4.0 if loose else 2.0) * amp, 5e-2):
            bad.append(report[-1])
    print("fused bilinear bwd rel err (ours, oracle under autocast):", report)
    assert not bad, (bad, report)
    # A/B against the materialised device dataflow on the same inputs
    m.load_state_dict(sd)
    out_b, g_b, used_b = run_dev(case, m, w, fused=False)
    assert used_b["fn"] != "_EmodPoolBackward"
    assert rel(out, out_b) < 2e-2, rel(out, out_b)
    assert rel(g[0], g_b[0]) < 1e-1, rel(g[0], g_b[0])


def test_fused_bilinear_is_deterministic():
    case = make_case(3, 2000, 64, ragged)
    w = torch.randn(2000, 64, generator=case["gen"])
    ref, m = build(case, 64, 4, True)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    out1, g1, _ = run_dev(case, m, w, fused=True)
    m.load_state_dict(sd)
    out2, g2, _ = run_dev(case, m, w, fused=True)
    assert torch.equal(out1, out2)
    assert torch.equal(g1[0], g2[0])            # the feature-map gradient: segmented reductions, no atomics


def test_fused_bilinear_through_the_data_objects():
    """get_mapped_features(interpolate=True) of an exact mapping stays lazy and GroupBimodalCSRPool takes the fused path."""
    from deepviewagg_amd import ops
    from deepviewagg_amd.core.multimodal.image import ImageMapping, SameSettingImageData
    from deepviewagg_amd.modules.multimodal import pooling as P
    gen = torch.Generator().manual_seed(0)
    B, C, H, W, N = 2, 64, 8, 16, 300
    pts = torch.arange(N).repeat_interleave(2)
    imgs = torch.arange(2).repeat(N)
    pix = torch.stack([torch.randint(0, W * 4, (2 * N,), generator=gen), torch.randint(0, H * 4, (2 * N,), generator=gen)], 1)
    feats = torch.rand(2 * N, 8, generator=gen)
    m = ImageMapping.from_dense(pts.to(DEV), imgs.to(DEV), pix.short().to(DEV), feats.to(DEV), num_points=N)
    sd = SameSettingImageData(path=[f"i{i}" for i in range(B)], pos=torch.rand(B, 3), opk=torch.zeros(B, 3),
                              ref_size=(W * 4, H * 4), proj_upscale=1, mappings=m,
                              x=torch.zeros(B, 3, H * 4, W * 4, dtype=torch.uint8)).to(DEV)
    x = torch.randn(B, C, H, W, generator=gen).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    sd.x = x.requires_grad_()
    pool = P.GroupBimodalCSRPool(in_map=8, in_mod=C, num_groups=4, use_num=True).to(DEV).train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xm = sd.get_mapped_features(interpolate=True)
        assert isinstance(xm, ops.InterpolatedFeatures)
        xm = P.BimodalCSRPool(mode='max')(None, xm, None, sd.atomic_csr_indexing)
        out = pool(None, xm, sd.mapping_features, sd.view_csr_indexing)
    assert type(out.grad_fn).__name__ == "_EmodPoolBackward" and out.shape == (N, C)
    out.float().sum().backward()
    assert sd.x.grad is not None and torch.isfinite(sd.x.grad.float()).all()


@pytest.mark.parametrize("train", [True, False])
def test_materialised_fallback_hoists_the_first_linear(train):
    """What the fused kernels do not cover (here: C_out = 256 with two channel groups; the published 512 -> 256 level has
    four and trains on them since round 4) takes the materialised fallback: it runs E_mod's first Linear on the map rows and interpolates its C_out
    channels (interp(x) W^T = interp(x W^T)) instead of materialising [V, C_in] and a per-view GEMM.  Checked against
    the oracle like the fused path, and against the un-hoisted device dataflow."""
    from deepviewagg_amd.modules.multimodal import pooling as P
    CO, G_ = 256, 2          # (C_out = 256 is fused for G = 4 only)
    case = make_case(31, 1200, 96, ragged)
    ref, m = build(case, CO, G_, train)
    w = torch.randn(case["N"], CO, generator=case["gen"])
    calls = []
    orig = P._hoisted_first_linear

    def spy(mlp, x_mod):
        r = orig(mlp, x_mod)
        calls.append(r[1])
        return r
    P._hoisted_first_linear = spy
    try:
        out, g, used = run_dev(case, m, w, fused=True)
        assert calls == [True] and used["fn"] != "_EmodPoolBackward"
        P._hoisted_first_linear = lambda mlp, x_mod: (x_mod.materialize(), False)
        _, m2 = build(case, CO, G_, train)
        out_b, g_b, _ = run_dev(case, m2, w, fused=True)
    finally:
        P._hoisted_first_linear = orig
    out_ref, g_ref = oracle(case, ref, w, autocast=False)
    out_amp, _ = oracle(case, ref, w, autocast=True)
    r, r_amp = rel(out, out_ref), rel(out_amp, out_ref)
    assert r < max(2e-2, 1.5 * r_amp), (r, r_amp)
    assert rel(out, out_b) < 2e-2, rel(out, out_b)
    assert rel(g[0], g_b[0]) < 1e-1, rel(g[0], g_b[0])                    # feature-map gradient (two bf16 dataflows)
    assert rel(g[0], g_ref[0]) < max(5e-2, 2 * rel(g_b[0], g_ref[0]))


@pytest.mark.parametrize("C", [32, 64])
def test_emod_bwd_stage1_in_place_batchnorm_backward(C):
    """dva_emod_bwd(stage 1) through the C ABI: da <- G dy_a - K1 - K2 z_a in place from the stored z_a (the host path
    folds this into the anchor scatter since round 3; the entry stays in the ABI)."""
    from deepviewagg_amd import _lib, fused_chain, fused_bilinear
    from deepviewagg_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    gen = torch.Generator().manual_seed(100 + C)
    sizes = ragged_long(700, gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(DEV)
    V = int(csr[-1])
    tiles, n_tiles = fused_chain.build_tiles(csr, V)
    z_a = torch.randn(V, C, generator=gen).bfloat16().to(DEV)
    dy = torch.randn(V, C, generator=gen).bfloat16().to(DEV)
    bn = torch.stack([torch.randn(C, generator=gen) * 0.2, torch.rand(C, generator=gen) + 0.5,
                      torch.randn(C, generator=gen) * 0.3 + 1.0, torch.randn(C, generator=gen)]).to(DEV).contiguous()
    sm = (torch.randn(2 * C, generator=gen) * 0.05).to(DEV)
    kappa = fused_bilinear.position_order(C, DEV)
    mean, inv, gam = bn[0][kappa], bn[1][kappa], bn[2][kappa]
    s1, s2 = sm[:C][kappa], sm[C:][kappa]
    g = gam * inv
    ref = (g * dy.float() - g * (s1 - mean * inv * s2) - g * inv * s2 * z_a.float())
    da = dy.clone()
    check(lib.dva_emod_bwd(1, None, None, None, ptr(tiles), ptr(n_tiles), None, ptr(bn), None, ptr(sm), None, None, None,
                           ptr(da), None, None, ptr(z_a), 700, V, 1, C, 1, stream_of(da)), "dva_emod_bwd")
    torch.testing.assert_close(da.float(), ref, rtol=1e-2, atol=1e-2)          # bf16 output
    assert float((da.float() - ref).norm() / ref.norm()) < 4e-3


@pytest.mark.parametrize("sizes_fn,N,C_in,G,C_out", [(ragged, 1500, 256, 4, 128), (ragged_long, 800, 96, 2, 128),
                                                     (full32, 200, 256, 1, 128), (ragged_long, 900, 160, 4, 256)])
def test_fused_bilinear_eval_c128(sizes_fn, N, C_in, G, C_out):
    """C_out = 128 / 256 (the KITTI-360 pyramid levels 256 -> 128, 512 -> 256): eval mode under no_grad runs the ONE fused
    kernel on the taps of Y (block by block since round 4: 183 - 203 registers at 128, no spills at 256).  With grad
    enabled C_out = 128 stays on the fused path (its backward kernels exist since round 4), C_out = 256 takes the
    materialised fallback with the hoisted Linear_a."""
    from deepviewagg_amd import fused_bilinear
    from deepviewagg_amd.modules.multimodal import pooling as P
    case = make_case(41 + G, N, C_in, sizes_fn)
    ref, m = build(case, C_out, G, train=False)
    w = torch.randn(case["N"], C_out, generator=case["gen"])
    calls = []
    orig = fused_bilinear.pool

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    fused_bilinear.pool = spy
    try:
        with torch.no_grad():
            out, _, _ = run_dev(case, m, w, fused=True, need_grad=False)
        assert calls == [1], "the fused eval kernel must be the path that ran"
        out_grad, _, used = run_dev(case, m, w, fused=True, need_grad=True)       # grad enabled
        assert calls == [1, 1] and used["fn"] == "_EmodPoolBackward"      # both widths train on the fused path (round 4)
    finally:
        fused_bilinear.pool = orig
    with torch.no_grad():
        xm = O.gather_bilinear(case["x"], case["images"], case["pixels"], case["msize"])
        out_ref = ref(None, xm, case["x_map"], case["csr"])
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out_amp = ref(None, xm, case["x_map"], case["csr"])
    r, r_amp = rel(out, out_ref), rel(out_amp, out_ref)
    assert out.dtype == torch.bfloat16 and r < max(2e-2, 1.5 * r_amp), (r, r_amp)
    assert rel(out, out_grad) < 2e-2
    unseen = (case["csr"][1:] == case["csr"][:-1])
    assert float(out.float().cpu()[unseen].abs().max() if unseen.any() else 0.0) == 0.0


# ---------------------------------------------------------------------------------------------------------------
# The fused bilinear path against a fixture written by the REFERENCE's own sparse_interpolation +
# GroupBimodalCSRPool (oracle/gen_golden.py pools_bilinear; core/multimodal/image.py:105-170,1278-1283,
# modules/multimodal/pooling.py:263-315): C_in 128 -> C_o 32, G = 4, points with 32 / 40 / 70 views, unseen points,
# border pixels, train and eval mode.  Gates as in test_chain_against_reference_fixture.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["pool_group_bilinear_train", "pool_group_bilinear_eval"])
def test_fused_bilinear_against_reference_fixture(name):
    import ast
    from conftest import load_golden, t, state_dict_from
    from deepviewagg_amd.modules.multimodal import pooling as P
    g = load_golden(name)
    kwargs = ast.literal_eval(str(g["kwargs"]))
    train = bool(g["train"])
    msize = tuple(int(v) for v in g["mapping_size"])
    case = dict(csr=t(g["csr"]), V=int(g["csr"][-1]), images=t(g["images"]), pixels=t(g["pixels"]), x=t(g["x"]),
                x_map=t(g["x_map"]), N=len(g["csr"]) - 1, C=kwargs["in_mod"], msize=msize)
    w = t(g["w"])
    m = P.GroupBimodalCSRPool(**kwargs)
    m.load_state_dict(state_dict_from(g), strict=True)
    m = m.to(DEV).train(train)
    out, grads, used = run_dev(case, m, w, fused=True)
    assert used["fn"] == "_EmodPoolBackward", f"the fused bilinear path must be the one that ran ({used})"
    # yardstick: the oracle under autocast against the same fixture
    ref = O.GroupBimodalCSRPool(**kwargs)
    ref.load_state_dict(state_dict_from(g), strict=True)
    ref.train(train)
    out_amp, g_amp = oracle(case, ref, w, autocast=True)
    r, r_amp = rel(out, t(g["out"])), rel(out_amp, t(g["out"]))
    print(f"fused bilinear vs reference fixture {name}: out {r:.4f} (oracle under autocast {r_amp:.4f})")
    assert r < max(2e-2, 1.5 * r_amp), (r, r_amp)
    unseen = case["csr"][1:] == case["csr"][:-1]
    assert float(out.detach().float().cpu()[unseen].abs().max()) == 0.0
    names = ["x"] + [n for n, _ in m.named_parameters()]
    refs = [t(g["grad_x"])] + [t(g["gp/" + n]) for n in names[1:]]
    amps = sorted(rel(c, b) for n, b, c in zip(names, refs, g_amp)
                  if c is not None and (n.startswith("E_map") or n.startswith("E_mod")))
    med = amps[len(amps) // 2]
    bad, report = [], []
    for n, a, b, c in zip(names, grads, refs, g_amp):
        assert a is not None, f"no gradient for {n}"
        if float(b.abs().max()) == 0:
            assert float(a.abs().max()) == 0, n
            continue
        ours, amp = rel(a, b), (rel(c, b) if c is not None else 0.0)
        report.append((n, round(ours, 4), round(amp, 4)))
        if n.startswith("E_map") or n.startswith("E_mod"):
            amp = max(amp, med)
        loose = n.startswith("G.") or n.startswith("E_score")
        if ours > max((4.0 if loose else 2.0) * amp, 5e-2):
            bad.append(report[-1])
    print("fused bilinear vs reference fixture, gradients (ours, oracle under autocast):", report)
    assert not bad, (bad, report)
    if train:
        for k, v in m.state_dict().items():
            if "running" in k:
                torch.testing.assert_close(v.cpu(), t(g["sd_after/" + k]), rtol=2e-2, atol=2e-3)


def test_anchor_scatter_workspace_chunks_and_gram_form():
    """(a) the per-anchor workspace of ``ops.bilinear_scatter`` is bounded -- with a budget of one image per
    pass the map gradient is bit-identical to the one-pass result; (b) the BatchNorm_a backward at the level of the
    anchor (Gram matrix of the tap weights x unrounded interpolation of Y, ``DVA_ANCHOR_GRAM=1``, the default) against
    the row-by-row form on the stored bf16 z_a: equal to bf16 rounding of z_a (2^-9 relative per value)."""
    from deepviewagg_amd import ops, fused_bilinear
    case = make_case(9, 1800, 64, ragged, B=4)
    w = torch.randn(1800, 64, generator=case["gen"])
    _, m = build(case, 64, 4, True)
    sd = {k: v.clone() for k, v in m.state_dict().items()}

    def gx(budget, gram):
        m.load_state_dict(sd)
        ops.ANCHOR_WS_BYTES, old = budget, fused_bilinear.ANCHOR_GRAM
        fused_bilinear.ANCHOR_GRAM = gram
        try:
            _, g, used = run_dev(case, m, w, fused=True)
        finally:
            ops.ANCHOR_WS_BYTES, fused_bilinear.ANCHOR_GRAM = None, old
        assert used["fn"] == "_EmodPoolBackward"
        return g[0]
    g_one = gx(None, True)
    g_chunks = gx(13 * 21 * 4 * 64 * 4 * 1 + 8, True)        # one image of (H + 1)(W + 1) anchors per pass
    assert torch.equal(g_one, g_chunks)
    g_rows = gx(None, False)
    assert rel(g_one, g_rows) < 5e-3, rel(g_one, g_rows)
    # the materialised gather's backward goes through the same scatter (fp32 and bf16 rows)
    for dt in (torch.float32, torch.bfloat16):
        x = case["x"].to(DEV).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_()
        V = case["V"]
        packed = ops.pack_gather_index(case["images"].to(DEV), torch.arange(V + 1, device=DEV), case["pixels"].to(DEV))
        res = torch.tensor([case["msize"]], dtype=torch.float32, device=DEV)
        coords = (case["pixels"].to(DEV) / (res - 1))[:, [1, 0]]
        go = torch.randn(V, 64, device=DEV, dtype=dt)
        outs = []
        for budget in (None, 13 * 21 * 4 * 64 * 4 * 2):
            ops.ANCHOR_WS_BYTES = budget
            try:
                (g_,) = torch.autograd.grad(ops.gather_bilinear(x, packed, coords), x, go)
            finally:
                ops.ANCHOR_WS_BYTES = None
            outs.append(g_)
        assert torch.equal(outs[0], outs[1])


# ---------------------------------------------------------------------------------------------------------------------
# Tight parity of the fused bilinear path, the WIDE levels included (round 5): the bf16-emulation oracle
# of the recompute chain (oracle/chain_emulation.py, tests/test_gpu_chain.py) extended by E_mod's roundings -- Y = Linear_a
# on the map rows stored as bf16, z_a stored as bf16 in train mode with BatchNorm_a's statistics taken from the stored
# values, bf16 operands of Linear_b -- and held to the FIXED tolerances of test_gpu_chain.EMU_TOL (relative L2 per tensor)
# instead of the autocast-relative gate above.  As there, the discrete decisions (rounding of the chain's folded
# operands, arg-max views / gate branches) are shared with the device: dev_invstd, dev_scores.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sizes_fn,N,C_in,C_out,G,train", [
    (ragged_long, 1200, 256, 128, 4, True),     # KITTI-360 pyramid level 256 -> 128 (block-by-block kernels)
    (full32, 2048, 64, 128, 2, True),           # (sizes as in test_gpu_chain's emulation cases: on a few hundred points
    (ragged_long, 1500, 160, 256, 4, True),     #  the train-mode encoder gradients are chaotic at the 10 % level --
    (full32, 1024, 64, 256, 4, True),           #  tests/test_oracle_chaos.py; 512 points, G = 1 measured 11 % on E_map
                                                #  with every E_mod gradient at 0.2 - 0.6 %)
    (ragged_long, 1200, 96, 256, 4, False),
    (ragged, 3000, 64, 64, 4, True),            # the register-resident widths for comparison
    (ragged_long, 1500, 128, 32, 4, True),
])
def test_fused_bilinear_matches_bf16_emulation(sizes_fn, N, C_in, C_out, G, train):
    from deepviewagg_amd import ops
    from deepviewagg_amd.modules.multimodal import pooling as P
    from oracle.chain_emulation import emulated_chain, emulated_emod
    from test_gpu_chain import EMU_TOL
    case = make_case(33, N, C_in, sizes_fn)
    w = torch.randn(N, C_out, generator=case["gen"])
    ref, m = build(case, C_out, G, train)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    V, csr = case["V"], case["csr"]
    # ---- device: forward, the saved BatchNorm tables / scores, then the gradients
    xd = case["x"].to(DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    atom_ptr = torch.arange(V + 1, device=DEV)
    packed = ops.pack_gather_index(case["images"].to(DEV), atom_ptr, case["pixels"].to(DEV))
    res = torch.tensor([case["msize"]], dtype=torch.float32, device=DEV)
    coords = (case["pixels"].to(DEV) / (res - 1))[:, [1, 0]]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lazy = ops.lazy_gather_bilinear(xd, packed, coords, exact=True)
        lazy = P.BimodalCSRPool(mode='max')(None, lazy, None, atom_ptr)
        out = m(None, lazy, case["x_map"].to(DEV), csr.to(DEV))
    assert type(out.grad_fn).__name__ == "_EmodPoolBackward", "the fused bilinear path must be the one that ran"
    saved = out.grad_fn.saved_tensors          # fused_bilinear._EmodPool.forward: ... bn1 [13], bn2, bn5, bn6, out, scores [18]
    dev_invstd = {1: saved[13][1].cpu(), 2: saved[14][1].cpu(), 6: saved[16][1].cpu()}
    dev_scores = saved[18].cpu()[:, :G]
    g = torch.autograd.grad((out.float() * w.to(DEV)).sum(), [xd] + list(m.parameters()), allow_unused=True)
    # ---- emulation: own forward (output, scores), then the gradient leg at the device's scores
    names = ["x"] + [n for n, _ in ref.named_parameters()]
    with torch.no_grad():
        vals = emulated_emod(ref, case["x"], case["images"], case["pixels"], case["msize"])
        out_own, sc_own = emulated_chain(ref, vals, case["x_map"], csr, dev_invstd=dev_invstd, return_scores=True)
    ref.load_state_dict(sd)
    xr = case["x"].clone().requires_grad_()
    # (a backward behind an EVAL forward differentiates the evaluation from the stored bf16 z_a: stored_za=True)
    vals = emulated_emod(ref, xr, case["images"], case["pixels"], case["msize"], stored_za=True)
    out_ref = emulated_chain(ref, vals, case["x_map"], csr, dev_invstd=dev_invstd, dev_scores=dev_scores)
    g_ref = torch.autograd.grad((out_ref * w).sum(), [xr] + list(ref.parameters()), allow_unused=True)
    r_out, r_sc = rel(out, out_own), rel(dev_scores, sc_own)
    report, bad, par = [("out", round(r_out, 5)), ("scores", round(r_sc, 5))], [], []
    for n, a, b in zip(names, g, g_ref):
        if b is None:
            assert a is None or float(a.abs().max()) == 0, n
            continue
        assert a is not None, n
        r = rel(a, b)
        if n == "E_score.bias":
            # sum of the score gradients: zero up to the gate's share (softmax is shift invariant) -- measured against
            # the scale of the score layer's gradient, not against its own near-zero norm
            scale = float(g_ref[names.index("E_score.weight")].norm())
            r = float((a.float().cpu() - b).norm()) / max(float(b.norm()), 1e-2 * scale)
        report.append((n, round(r, 5)))
        if n != "x":
            par.append(r)
        # what the bilinear kernels compute (feature-map gradient, E_mod, score layer, gate) at the fixed tolerances; the
        # mapping-feature encoder is the SHARED chain code (held to them by test_gpu_chain.py): its set-branch tensors sit
        # behind the arg-max of the set pooling and reach 5 % on the eval case here (2 % on test_gpu_chain's): train gate
        tol = EMU_TOL["rows"] if n == "x" else EMU_TOL["param_train" if (train or n.startswith("E_map")) else "param_eval"]
        if r > tol:
            bad.append((n, r))
    print("fused bilinear vs bf16 emulation, rel L2:", report)
    import os
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/emu_report_bilinear_r5.txt", "a") as f:
            f.write(f"{sizes_fn.__name__} N={N} C_in={C_in} C_out={C_out} G={G} train={train} {report}\n")
    assert r_out < EMU_TOL["out"], report
    assert r_sc < EMU_TOL["scores"], report
    assert rel(out, out_ref) < 2 * EMU_TOL["out"] if not train else rel(out, out_ref) < EMU_TOL["out"], report
    assert not bad, (bad, report)
    if train:
        assert sorted(par)[len(par) // 2] < EMU_TOL["param_train_median"], report


def test_interpolated_features_cat_equals_torch_composition():
    """ops.InterpolatedFeatures.cat (dva_bilinear_taps_cat): the taps of three settings concatenated, offset into the stacked
    row / anchor numbering and permuted -- against the same thing spelled with torch ops; materialize() of the result =
    the reference's dataflow (cat of the settings' [V_s, C] gathers, indexed by the order)."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(4)
    geo = [(3, 12, 20), (2, 7, 9), (4, 16, 8)]
    items = []
    for (B, H, W), P in zip(geo, (5000, 1, 3000)):
        x = torch.randn(B, 16, H, W, generator=gen).to(DEV)
        images = torch.randint(0, B, (P,), generator=gen)
        pixels = torch.stack([torch.randint(0, 4 * W, (P,), generator=gen), torch.randint(0, 4 * H, (P,), generator=gen)], 1)
        packed = ops.pack_gather_index(images.to(DEV), torch.arange(P + 1, device=DEV), pixels.short().to(DEV), ratio=1.0)
        res = torch.tensor([[4 * W, 4 * H]], dtype=torch.float32)
        coords = (pixels / (res - 1))[:, [1, 0]].contiguous().to(DEV)
        # (q + 1 rounding across an integer: a few views get the dummy anchor)
        items.append(ops.lazy_gather_bilinear(x, packed, coords, True))
    V = sum(it.shape[0] for it in items)
    order = torch.randperm(V, generator=gen).to(DEV)
    for od in (order, None):
        cat = ops.InterpolatedFeatures.cat(items, order=od)
        dummy = ops.n_anchors(geo)
        r_off = a_off = 0
        rows4, anchors = [], []
        for it, g in zip(items, geo):
            na = ops.n_anchors([g])
            rows4.append(it.tap_rows + r_off)
            anchors.append(torch.where(it.anchors == na, torch.full_like(it.anchors, dummy), it.anchors + a_off))
            r_off += it.rows.shape[0]
            a_off += na
        rows4, w4, anchors = torch.cat(rows4), torch.cat([it.tap_weights for it in items]), torch.cat(anchors)
        if od is not None:
            rows4, w4, anchors = rows4[od], w4[od], anchors[od]
        assert torch.equal(cat.tap_rows, rows4) and torch.equal(cat.tap_weights, w4) and torch.equal(cat.anchors, anchors)
        assert cat.geometry == [tuple(g) for g in geo] and cat.rows.shape[0] == r_off and cat.shape == (V, 16)
        want = torch.cat([it.materialize() for it in items])
        assert torch.equal(cat.materialize(), want if od is None else want[od])
        # the taps reproduce the gather on the stacked rows
        via_taps = (cat.rows[cat.tap_rows.long()].float() * cat.tap_weights.unsqueeze(-1)).sum(1)
        torch.testing.assert_close(via_taps, cat.materialize().float(), rtol=1e-5, atol=1e-5)
    assert ops.InterpolatedFeatures.cat(items[:1]) is items[0]


@pytest.mark.parametrize("env", [{"DVA_EMOD_COOP2": "1"}, {"DVA_EMOD_COOP_DYA": "0"}])
def test_c256_backward_variants_stay_green(env):
    """The A/B forms of the C_o = 256 E_mod backward (round 6: `emodw_coop2_kernel` = no dz_b hand-off; the round-4 pair
    emodw_wgrad_coop + MODE 4) are selected by switches the library reads once per process: the C_o = 256 oracle /
    emulation cases of this file run again in a child process under each switch."""
    import os
    import subprocess
    import sys
    if os.environ.get("DVA_BILINEAR_VARIANT_CHILD") == "1":
        pytest.skip("child process")
    e = dict(os.environ, DVA_BILINEAR_VARIANT_CHILD="1", **env)
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-m", "gpu", "-q", "-x", "-k", "256 and not variants"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(here)))
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
