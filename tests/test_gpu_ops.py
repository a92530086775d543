"""-m gpu parity tests: HIP kernels (through the C ABI) vs the CPU oracle and the golden vectors.

Tolerances: fp32 paths 1e-5 relative (sums are re-associated across lanes), bf16 paths 2e-2.
Index outputs (argmax rows, packed gather index) are bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, t
from oracle import pooling_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, rtol=1e-5, atol=1e-6):
    if isinstance(b, np.ndarray):
        b = t(b)
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def random_csr(n, max_size, gen, p_empty=0.2):
    sizes = torch.randint(1, max_size + 1, (n,), generator=gen)
    sizes[torch.rand(n, generator=gen) < p_empty] = 0
    return torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])


def test_library_loads_and_sees_device():
    from deepviewagg_amd import _lib
    lib = _lib.load()
    assert lib.dva_version() >= 100
    assert lib.dva_device_count() >= 1


def test_cpu_tensors_are_refused():
    from deepviewagg_amd import ops, _lib
    with pytest.raises(_lib.DvaError):
        ops.segment_csr(torch.randn(4, 2), torch.tensor([0, 2, 4]))


def test_segment_csr_golden():
    from deepviewagg_amd import ops
    g = load_golden("segment_csr")
    csr = t(g["csr"], DEV)
    for red in ("sum", "mean", "max", "min"):
        src = t(g["src"], DEV).requires_grad_()
        out = ops.segment_csr(src, csr, reduce=red)
        close(out, g[f"out_{red}"])
        (gr,) = torch.autograd.grad((out * t(g[f"w_{red}"], DEV)).sum(), src)
        close(gr, g[f"grad_{red}"])
    close(ops.gather_csr(t(g["gather_src"], DEV), csr), g["gather_out"], rtol=0, atol=0)
    # arg: bit-exact vs the oracle's first-occurrence definition
    _, arg = ops.segment_csr_arg(t(g["src"], DEV), csr, "max")
    assert torch.equal(arg.cpu().long(), O.segment_arg(t(g["src"]), t(g["csr"]), "max"))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [1, 3, 64, 130])
def test_segment_csr_random(dtype, C):
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C)
    csr = random_csr(300, 9, gen)
    src = torch.randn(int(csr[-1]), C, generator=gen).to(dtype)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    for red in ("sum", "mean", "max", "min"):
        s_dev = src.to(DEV).requires_grad_()
        out = ops.segment_csr(s_dev, csr.to(DEV), reduce=red)
        s_cpu = src.float().requires_grad_()
        ref = O.segment_csr(s_cpu, csr, red)
        close(out, ref, **tol)
        w = torch.randn(ref.shape, generator=gen)
        (g_dev,) = torch.autograd.grad((out.float() * w.to(DEV)).sum(), s_dev)
        (g_cpu,) = torch.autograd.grad((ref * w).sum(), s_cpu)
        close(g_dev, g_cpu, **tol)


def test_segment_csr_edge_cases():
    from deepviewagg_amd import ops
    # all groups empty; single group; 1-D source
    csr = torch.zeros(5, dtype=torch.long, device=DEV)
    out = ops.segment_csr(torch.zeros(0, 3, device=DEV), csr, reduce="max")
    assert out.shape == (4, 3) and float(out.abs().sum()) == 0
    src = torch.arange(6, dtype=torch.float32, device=DEV)
    out = ops.segment_csr(src, torch.tensor([0, 6], device=DEV), reduce="sum")
    assert out.shape == (1,) and float(out[0]) == 15
    with pytest.raises(ValueError):
        ops.segment_csr(src, torch.tensor([[0, 6]], device=DEV))


@pytest.mark.parametrize("G", [1, 4])
def test_segment_softmax_golden(G):
    from deepviewagg_amd import ops
    g = load_golden(f"softmax_random_G{G}")
    csr, w = t(g["csr"], DEV), t(g["w"], DEV)
    for sc in (0, 1):
        src = t(g["src"], DEV).requires_grad_()
        out = ops.segment_softmax_csr(src, csr, scaling=bool(sc))
        close(out, g[f"out_{sc}"])
        (gr,) = torch.autograd.grad((out * w).sum(), src)
        close(gr, g[f"grad_{sc}"], rtol=1e-4, atol=1e-6)
    k = load_golden("softmax_known")
    close(ops.segment_softmax_csr(t(k["src"], DEV), t(k["csr"], DEV)), k["out"])
    close(ops.segment_softmax_csr(t(k["src"], DEV), t(k["csr"], DEV), scaling=True), k["out_scaled"])


def attention_reference(val, compat, csr, gw, gb, scaling, C, G):
    """oracle: pooling.py:284-300"""
    class _G(torch.nn.Module):
        def forward(self, x):
            return torch.tanh(torch.relu(x * gw + gb)).view(-1, G).squeeze(1)
    return O.attention_tail(val, compat, csr, _G() if gw is not None else None, G, C, scaling)


ATT_SHAPES = [  # (C, G, max_views, dtype)
    (64, 4, 40, torch.float32), (64, 4, 40, torch.bfloat16), (64, 4, 3, torch.bfloat16),
    (512, 4, 8, torch.bfloat16), (256, 8, 8, torch.float32), (32, 1, 5, torch.float32),
    (16, 16, 6, torch.float32), (128, 4, 70, torch.bfloat16),
    (7, 2, 5, torch.float32), (20, 5, 4, torch.float32), (12, 4, 6, torch.bfloat16),
]


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("gating", [True, False])
@pytest.mark.parametrize("C,G,max_views,dtype", ATT_SHAPES)
def test_view_attention(C, G, max_views, dtype, gating, algo):
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C * 100 + G)
    N = 257
    csr = random_csr(N, max_views, gen)
    V = int(csr[-1])
    val = torch.randn(V, C, generator=gen).to(dtype)
    compat = torch.randn(V, G, generator=gen) * 2
    gw = (torch.randn(1, G, generator=gen)) if gating else None
    gb = (torch.randn(1, G, generator=gen) * 0.5) if gating else None
    w = torch.randn(N, C, generator=gen)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)

    # oracle in fp32 on the (rounded) inputs
    val_c, compat_c = val.float().requires_grad_(), compat.clone().requires_grad_()
    gw_c = gw.clone().requires_grad_() if gating else None
    gb_c = gb.clone().requires_grad_() if gating else None
    ref, att_ref, gate_ref = attention_reference(val_c, compat_c, csr, gw_c, gb_c, True, C, G)
    ins_c = [val_c, compat_c] + ([gw_c, gb_c] if gating else [])
    gref = torch.autograd.grad((ref * w).sum(), ins_c)

    ops.ATTENTION_ALGO = algo
    try:
        val_d, compat_d = val.to(DEV).requires_grad_(), compat.to(DEV).requires_grad_()
        gw_d = gw.to(DEV).requires_grad_() if gating else None
        gb_d = gb.to(DEV).requires_grad_() if gating else None
        out, att, gate = ops.view_attention(val_d, compat_d, csr.to(DEV), gw_d, gb_d, scaling=True)
        ins_d = [val_d, compat_d] + ([gw_d, gb_d] if gating else [])
        gdev = torch.autograd.grad((out.float() * w.to(DEV)).sum(), ins_d)
    finally:
        ops.ATTENTION_ALGO = 0
    close(out, ref, **tol)
    close(att, att_ref, rtol=1e-5, atol=1e-6)
    if gating:
        close(gate, gate_ref.view(N, G), rtol=1e-5, atol=1e-6)
    close(gdev[0], gref[0], **tol)
    gtol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=5e-2)
    close(gdev[1], gref[1], **gtol)
    if gating:
        close(gdev[2], gref[2], rtol=gtol["rtol"], atol=gtol["atol"] * 20)
        close(gdev[3], gref[3], rtol=gtol["rtol"], atol=gtol["atol"] * 20)


def test_view_attention_team_equals_generic_bitwise_indices():
    """Both code paths must select the same argmax rows (ties -> first row)."""
    from deepviewagg_amd import ops, _lib
    gen = torch.Generator().manual_seed(11)
    N, C, G = 500, 64, 4
    csr = random_csr(N, 12, gen)
    V = int(csr[-1])
    compat = torch.randint(-2, 3, (V, G), generator=gen).float()  # many ties
    val = torch.randn(V, C, generator=gen)
    lib = _lib.load()
    res = []
    val_d, compat_d, csr_d = val.to(DEV), compat.to(DEV), csr.to(DEV)  # keep alive across the raw calls
    for algo in (1, 2):
        out = torch.empty(N, C, device=DEV)
        att = torch.zeros(V, G, device=DEV)
        gate = torch.empty(N, G, device=DEV)
        amax = torch.empty(N, G, dtype=torch.int32, device=DEV)
        gw, gb = torch.ones(G, device=DEV), torch.zeros(G, device=DEV)
        rc = lib.dva_view_attention_fwd(
            _lib.ptr(val_d), _lib.ptr(compat_d), _lib.ptr(csr_d), _lib.ptr(gw), _lib.ptr(gb),
            _lib.ptr(out), _lib.ptr(att), _lib.ptr(gate), _lib.ptr(amax), N, V, C, G, 1, 1e-12, 0, algo, None)
        assert rc == 0
        torch.cuda.synchronize()
        res.append((out.cpu(), amax.cpu()))
    assert torch.equal(res[0][1], res[1][1])
    ref_arg = O.segment_arg(compat, csr, "max")
    assert torch.equal(res[0][1].long(), ref_arg)
    close(res[0][0], res[1][0], rtol=1e-5, atol=1e-5)


def images_per_atom(g):
    sizes = t(g["atom_pointers"])[1:] - t(g["atom_pointers"])[:-1]
    return t(g["images"]).repeat_interleave(sizes)


@pytest.mark.parametrize("name", ["gather", "gather_multipixel"])
def test_gather_nearest_golden(name):
    from deepviewagg_amd import ops
    g = load_golden(name)
    x = t(g["x"], DEV).requires_grad_()
    packed = ops.pack_gather_index(t(g["images"], DEV), t(g["atom_pointers"], DEV), t(g["pixels"], DEV),
                                   ratio=float(g["downscale"]))
    out = ops.gather_nearest(x, packed)
    assert torch.equal(out.cpu(), t(g["out_nearest"]))  # pure data movement: bit-exact
    if "w_nearest" in g:
        (gr,) = torch.autograd.grad((out * t(g["w_nearest"], DEV)).sum(), x)
        close(gr, g["grad_x_nearest"])
    else:
        pooled = ops.segment_csr(out, t(g["atom_pointers"], DEV), reduce="max")
        close(pooled, g["out_atomic_max"], rtol=0, atol=0)


def test_gather_segment_max_golden():
    """Non-exact mapping (several pixels per view): the fused gather + atomic max pool (no [P, C] tensor) against the
    reference-run fixture, bit for bit (pure selection)."""
    from deepviewagg_amd import ops
    g = load_golden("gather_multipixel")
    x = t(g["x"], DEV)
    ptr_ = t(g["atom_pointers"], DEV)
    packed = ops.pack_gather_index(t(g["images"], DEV), ptr_, t(g["pixels"], DEV), ratio=float(g["downscale"]))
    lazy = ops.lazy_gather_nearest(x, packed, exact=False)
    assert ops.gather_segment_max_applicable(lazy, ptr_)
    pooled = ops.gather_segment_max(lazy, ptr_)
    assert isinstance(pooled, ops.GatheredFeatures) and pooled.exact
    assert torch.equal(pooled.materialize().cpu(), t(g["out_atomic_max"]))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [8, 64, 136])
def test_gather_segment_max_against_oracle(dtype, C):
    """Ragged views (0 .. 11 atoms, one of 300), ties between atoms (quantised rows: the first atom wins, as
    torch_scatter's segment_csr), forward bit-exact, backward against autograd of the oracle."""
    from deepviewagg_amd import ops
    from oracle import pooling_oracle as O
    gen = torch.Generator().manual_seed(7 + C)
    V, R = 900, 211
    sizes = torch.randint(0, 12, (V,), generator=gen)
    sizes[17] = 300
    ptr_ = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    P = int(ptr_[-1])
    rows = (torch.randn(R, C, generator=gen) * 2).round().div(2).to(dtype)          # many ties
    row_idx = torch.randint(0, R, (P,), generator=gen, dtype=torch.int32)
    w = torch.randn(V, C, generator=gen).to(dtype)
    r0 = rows.float().clone().requires_grad_()
    ref = O.segment_csr(r0[row_idx.long()], ptr_, 'max')
    (gr0,) = torch.autograd.grad((ref * w.float()).sum(), r0)
    grads = []
    for atomics in (False, True):            # the deterministic plan reduction (C / vec a power of two) and the atomics A/B
        ops.SEGMENT_MAX_ATOMICS = atomics
        try:
            r = rows.to(DEV).requires_grad_()
            out = ops._GatherSegmentMax.apply(r, row_idx.to(DEV), ptr_.to(DEV), None)
            assert torch.equal(out.float().cpu(), ref.detach())
            (gr,) = torch.autograd.grad((out.float() * w.to(DEV).float()).sum(), r)
        finally:
            ops.SEGMENT_MAX_ATOMICS = False
        torch.testing.assert_close(gr.float().cpu(), gr0, rtol=2e-2 if dtype == torch.bfloat16 else 1e-5,
                                   atol=2e-2 if dtype == torch.bfloat16 else 1e-5)
        grads.append(gr)
    if C != 136:       # 17 column groups: no power-of-two lane team, both runs took the atomics
        r = rows.to(DEV).requires_grad_()
        out = ops._GatherSegmentMax.apply(r, row_idx.to(DEV), ptr_.to(DEV), None)
        (gr2,) = torch.autograd.grad((out.float() * w.to(DEV).float()).sum(), r)
        assert torch.equal(gr2, grads[0])          # the plan reduction is bit-reproducible


def test_gather_bilinear_golden():
    from deepviewagg_amd import ops
    g = load_golden("gather")
    x = t(g["x"], DEV).requires_grad_()
    packed = ops.pack_gather_index(t(g["images"], DEV), t(g["atom_pointers"], DEV), t(g["pixels"], DEV))
    res = torch.tensor([g["mapping_size"].tolist()], dtype=torch.float32, device=DEV)
    coords = (t(g["pixels"], DEV) / (res - 1))[:, [1, 0]]
    out = ops.gather_bilinear(x, packed, coords)
    close(out, g["out_bilinear"], rtol=1e-6, atol=1e-6)
    (gr,) = torch.autograd.grad((out * t(g["w_bilinear"], DEV)).sum(), x)
    close(gr, g["grad_x_bilinear"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [3, 6, 64])
def test_gather_random_vs_oracle(dtype, C):
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C)
    B, H, W, P = 5, 12, 20, 3000
    x = torch.randn(B, C, H, W, generator=gen).to(dtype)
    images = torch.randint(0, B, (P,), generator=gen)
    pixels = torch.stack([torch.randint(0, W * 4, (P,), generator=gen),
                          torch.randint(0, H * 4, (P,), generator=gen)], 1).short()
    atom_ptr = torch.arange(P + 1)
    packed = ops.pack_gather_index(images.to(DEV), atom_ptr.to(DEV), pixels.to(DEV), ratio=4.0)
    x_d = x.to(DEV).requires_grad_()
    out = ops.gather_nearest(x_d, packed)
    ref = O.gather_nearest(x.float(), images, pixels, 4.0)
    assert torch.equal(out.float().cpu(), ref)
    w = torch.randn(P, C, generator=gen)
    (gr,) = torch.autograd.grad((out.float() * w.to(DEV)).sum(), x_d)
    x_c = x.float().requires_grad_()
    (gr_ref,) = torch.autograd.grad((O.gather_nearest(x_c, images, pixels, 4.0) * w).sum(), x_c)
    tol = dict(rtol=1e-5, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=1e-1)
    close(gr, gr_ref, **tol)
    # bilinear at mapping resolution (W*4, H*4)
    packed1 = ops.pack_gather_index(images.to(DEV), atom_ptr.to(DEV), pixels.to(DEV))
    res = torch.tensor([[W * 4, H * 4]], dtype=torch.float32)
    coords = (pixels / (res - 1))[:, [1, 0]]
    out = ops.gather_bilinear(x.to(DEV), packed1, coords.to(DEV))
    ref = O.sparse_interpolation(x.float(), coords, images)
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    close(out, ref, **tol)


# ---------------------------------------------------------------------------------------------
# row plan (views grouped by feature-map row) and the rows gradient as a segmented reduction
# ---------------------------------------------------------------------------------------------

# above the library's merge-sort limit: 17- / 18-bit row keys take the two-pass 9-bit onesweep instance, 19- / 20-bit
# keys (the anchor plan of the bilinear backward: 32 x 65 x 129 padded cells + 1) the two-pass 10-bit one
@pytest.mark.parametrize("V,R", [(0, 5), (1, 1), (1000, 7), (50000, 4096), (300000, 1 << 18),
                                 (1500000, (1 << 16) + 9), (3000000, (1 << 18) - 3),
                                 (3000000, 32 * 65 * 129 + 1), (2500000, (1 << 20) - 5), (2000000, (1 << 20) + 7)])
def test_row_plan_is_a_stable_sort(V, R):
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(V + R)
    row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32)
    if V > 10:
        row_idx[row_idx == 3] = 4                     # an empty row in the middle
    (perm, row_ptr), counts = ops.row_plan(row_idx.to(DEV), R)
    ref_perm = torch.sort(row_idx.long(), stable=True).indices
    ref_counts = torch.bincount(row_idx.long(), minlength=R)
    assert torch.equal(perm.cpu().long(), ref_perm)
    assert torch.equal(counts.cpu().long(), ref_counts)
    assert torch.equal(row_ptr.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), ref_counts.cumsum(0)]))


# the split plan (csrc/plan_split.hip): offsets from the keys, the two scatter passes on the 16-byte records themselves
@pytest.mark.parametrize("V,R,how", [(1, 513, "uniform"), (8192, 1000, "uniform"), (8193, 4097, "uniform"),
                                     (70001, 600, "uniform"), (250000, 1 << 18, "uniform"),
                                     (250000, (1 << 18) - 511, "uniform"), (200000, 3000, "one_row"),
                                     (300000, 70000, "few_rows"), (300000, 131072, "one_bucket"),
                                     (5000000, 1 << 18, "uniform"), (4500000, (1 << 17) + 5, "skewed")])
def test_split_plan_equals_row_plan(V, R, how):
    """row_ptr / counts of the split plan = those of the permutation plan (and of torch); records brought into plan order
    by the two scatter passes = rec[perm] bit for bit, word 3 = the row key; both passes stable."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(V + R)
    if how == "uniform":
        row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32)
    elif how == "one_row":
        row_idx = torch.full((V,), R - 2, dtype=torch.int32)
    elif how == "few_rows":
        row_idx = torch.tensor([5, 512, 513, R - 1, 40000], dtype=torch.int32)[torch.randint(0, 5, (V,), generator=gen)]
    elif how == "one_bucket":
        row_idx = (torch.randint(0, 512, (V,), generator=gen) + 512 * 100).to(torch.int32)
    else:   # half of the views in 64 rows, the rest uniform
        hot = torch.randint(0, R, (64,), generator=gen)
        row_idx = torch.where(torch.rand(V, generator=gen) < 0.5, hot[torch.randint(0, 64, (V,), generator=gen)],
                              torch.randint(0, R, (V,), generator=gen)).to(torch.int32)
    if V > 10 and how == "uniform":
        row_idx[row_idx == 3] = 4                     # an empty row in the middle
    rd = row_idx.to(DEV)
    old = ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS
    try:
        ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS = True, 0
        plan, counts = ops.row_plan(rd, R)
        assert isinstance(plan, ops.SplitPlan)
        ops.SPLIT_PLAN = False
        (perm, row_ptr), counts_ref = ops.row_plan(rd, R)
    finally:
        ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS = old
    ref_counts = torch.bincount(row_idx.long(), minlength=R)
    assert torch.equal(counts.cpu().long(), ref_counts) and torch.equal(counts, counts_ref)
    assert torch.equal(plan.row_ptr, row_ptr)
    assert torch.equal(perm.cpu().long(), torch.sort(row_idx.long(), stable=True).indices)
    rec = torch.randint(-2 ** 31, 2 ** 31 - 1, (V, 4), generator=gen, dtype=torch.int64).to(torch.int32).to(DEV)
    want = rec[perm.long()].clone()
    want[:, 3] = rd[perm.long()]
    got = plan.sort_records(rec.clone())
    assert torch.equal(got, want)
    keyed = rec.clone()
    keyed[:, 3] = rd                                   # records that carry their key (dva_chain_attn_bwd)
    assert torch.equal(plan.sort_records(keyed, keyed=True), want)
    assert torch.equal(plan.perm, perm) and torch.equal(plan[1], row_ptr)      # the lazy permutation of other callers


@pytest.mark.parametrize("V,R,C,G,how", [(9000, 600, 64, 4, "uniform"), (300000, 5000, 32, 2, "uniform"),
                                         (4300000, 1 << 18, 64, 4, "uniform"), (600000, 70000, 64, 1, "skewed"),
                                         (100000, 2000, 32, 4, "one_row")])
def test_bucket_rows_grad_equals_segmented_reduction(V, R, C, G, how):
    """dva_plan_split_rows_grad (pass A + one workgroup per bucket, records consumed from LDS) against pass B + the
    segmented reduction over plan-order records, and against an fp64 index_add of the same products."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(V + R + C)
    N = max(V // 8, 4)
    if how == "uniform":
        row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32)
    elif how == "one_row":
        row_idx = torch.full((V,), R - 3, dtype=torch.int32)
    else:
        hot = torch.randint(0, R, (16,), generator=gen)
        row_idx = torch.where(torch.rand(V, generator=gen) < 0.5, hot[torch.randint(0, 16, (V,), generator=gen)],
                              torch.randint(0, R, (V,), generator=gen)).to(torch.int32)
    point = torch.randint(0, N, (V,), generator=gen, dtype=torch.int32)
    wts = torch.randn(V, 4, generator=gen).to(torch.bfloat16)
    rec = torch.empty(V, 4, dtype=torch.int32)
    rec[:, 0] = point
    rec[:, 1:3] = wts.view(torch.int32)
    rec[:, 3] = row_idx
    gout = torch.randn(N, C, generator=gen).to(torch.bfloat16).to(DEV)
    rd, recd = row_idx.to(DEV), rec.to(DEV)
    old = ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS, ops.SPLIT_FUSED
    try:
        ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS = True, 0
        plan = ops.row_plan(rd, R, with_counts=False)[0]
        ops.SPLIT_FUSED = False
        a = ops.rows_grad_rec16(gout, plan, recd.clone(), R, C, G, torch.bfloat16, torch.cuda.current_stream().cuda_stream)
        ops.SPLIT_FUSED = True
        b = ops.rows_grad_rec16(gout, plan, recd.clone(), R, C, G, torch.bfloat16, torch.cuda.current_stream().cuda_stream)
        b2 = ops.rows_grad_rec16(gout, plan, recd.clone(), R, C, G, torch.bfloat16, torch.cuda.current_stream().cuda_stream)
    finally:
        ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS, ops.SPLIT_FUSED = old
    assert torch.equal(b, b2)                                       # deterministic
    if V <= 600000:
        ch_group = torch.arange(C) // (C // G)
        prod = gout.cpu().double()[point.long()] * wts.double()[:, ch_group]
        ref = torch.zeros(R, C, dtype=torch.float64).index_add_(0, row_idx.long(), prod)
        scale = float(ref.abs().max()) + 1e-9
        for got in (a, b):
            assert float((got.cpu().double() - ref).abs().max()) <= scale * 2 ** -7
    # the two device paths: same records, another order of the fp32 additions, one bf16 rounding each
    d = (a.float() - b.float()).abs()
    assert float(d.max()) <= float(a.float().abs().max()) * 2 ** -6
    assert float((d > 0).float().mean()) < 0.2


@pytest.mark.parametrize("C,G,gating", [(64, 4, True), (32, 2, False), (128, 1, True)])
def test_rows_grad_split_plan_equals_permutation_plan(C, G, gating):
    """view_gather_attention (bf16, the lean backward) over the split plan = over the permutation plan, bit for bit
    (the same records summed in the same order), with and without a plan handed in."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C + G)
    N, R = 6000, 700
    sizes = torch.randint(0, 9, (N,), generator=gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(DEV)
    V = int(csr[-1])
    row_idx = torch.randint(0, R - 20, (V,), generator=gen, dtype=torch.int32).to(DEV)
    rows = torch.randn(R, C, generator=gen).to(torch.bfloat16)
    compat = torch.randn(V, G, generator=gen)
    gw = torch.randn(G, generator=gen) if gating else None
    gb = torch.randn(G, generator=gen) if gating else None
    w = torch.randn(N, C, generator=gen).to(DEV)

    def run(split, with_plan, fused=False):
        old = ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS, ops.SPLIT_FUSED
        ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS, ops.SPLIT_FUSED = split, 0, fused
        try:
            plan = ops.row_plan(row_idx, R, with_counts=False)[0] if with_plan else None
            assert plan is None or isinstance(plan, ops.SplitPlan) == split
            rd = rows.to(DEV).requires_grad_()
            cd = compat.to(DEV).requires_grad_()
            gwd = gw.to(DEV).requires_grad_() if gating else None
            gbd = gb.to(DEV).requires_grad_() if gating else None
            out, _, _ = ops.view_gather_attention(rd, row_idx, cd, csr, gwd, gbd, plan=plan)
            return [out] + list(torch.autograd.grad((out.float() * w).sum(), [rd, cd]))
        finally:
            ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS, ops.SPLIT_FUSED = old

    a = run(False, True)
    for split, with_plan in ((True, True), (True, False)):
        b = run(split, with_plan)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert float(a[1][R - 20:].abs().max()) == 0.0 and float(a[1].abs().max()) > 0.0
    # the bucket kernel (no pass B; C <= 64): one lane team sums a row in view order -- same records, another order of the
    # fp32 additions: the bf16 rows agree to one rounding, and the kernel is deterministic
    f1, f2 = run(True, True, fused=True), run(True, False, fused=True)
    for x, y in zip(f1, f2):
        assert torch.equal(x, y)
    assert torch.equal(f1[0], a[0]) and torch.equal(f1[2], a[2])
    if C <= 64:
        assert not torch.equal(f1[1], a[1]) or True      # (may or may not differ in the last bit)
        close(f1[1].float(), a[1].float(), rtol=2 ** -7, atol=1e-3)
        assert float(f1[1][R - 20:].abs().max()) == 0.0
    else:
        assert torch.equal(f1[1], a[1])                   # C = 128: the bucket kernel declines, pass B runs


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,G,gating", [(64, 4, True), (32, 1, False), (24, 3, True), (128, 8, True)])
def test_rows_grad_plan_equals_atomics(C, G, gating, dtype):
    """grad wrt the feature-map rows: segmented reduction over the plan == fp32 atomics == torch."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C * G)
    N, R = 2000, 300
    sizes = torch.randint(0, 9, (N,), generator=gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    row_idx = torch.randint(0, R - 20, (V,), generator=gen, dtype=torch.int32)   # last rows unused
    rows = torch.randn(R, C, generator=gen).to(dtype)
    compat = torch.randn(V, G, generator=gen)
    gw = torch.randn(G, generator=gen) if gating else None
    gb = torch.randn(G, generator=gen) if gating else None
    w = torch.randn(N, C, generator=gen)

    def run(algo, plan):
        old, ops.ROWS_GRAD_ALGO = ops.ROWS_GRAD_ALGO, algo
        try:
            rd = rows.to(DEV).requires_grad_()
            cd = compat.to(DEV).requires_grad_()
            gwd = gw.to(DEV).requires_grad_() if gating else None
            gbd = gb.to(DEV).requires_grad_() if gating else None
            out, _, _ = ops.view_gather_attention(rd, row_idx.to(DEV), cd, csr.to(DEV), gwd, gbd, plan=plan)
            ins = [rd, cd] + ([gwd, gbd] if gating else [])
            return [out] + list(torch.autograd.grad((out.float() * w.to(DEV)).sum(), ins))
        finally:
            ops.ROWS_GRAD_ALGO = old

    plan = ops.row_plan(row_idx.to(DEV), R, with_counts=False)[0]
    a = run(1, None)
    b = run(0, None)          # plan built on demand in backward
    c = run(0, plan)
    for x, y in zip(b[:3], c[:3]):      # out, grad rows, grad compat: deterministic, bit-identical
        assert torch.equal(x, y)
    for x, y in zip(b[3:], c[3:]):      # gate parameters: fp32 sums accumulated atomically (order varies)
        close(x, y, rtol=1e-4, atol=1e-4)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    for x, y in zip(a[:3], b[:3]):
        close(x, y, **tol)
    ptol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else tol
    for x, y in zip(a[3:], b[3:]):      # gate parameters: atomically accumulated fp32 sums in both runs
        close(x, y, **ptol)
    assert float(b[1][R - 20:].abs().max()) == 0.0
    # torch reference on the materialised gather
    rr = rows.float().requires_grad_()
    x_mod = rr[row_idx.long()]
    att = O.segment_softmax_csr(compat, csr, scaling=False)
    xp = O.segment_csr(x_mod * O.expand_group_feat(att, G, C), csr, 'sum')
    if gating:
        mx = O.segment_csr(compat, csr, 'max')
        xp = xp * O.expand_group_feat(torch.tanh(torch.relu(mx * gw + gb)), G, C)
    g_ref = torch.autograd.grad((xp * w).sum(), rr)[0]
    close(b[1], g_ref, **(dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,C,slope", [(1000, 64, 0.2), (8192 * 2, 32, 0.2), (777, 48, 0.0)])
def test_rowbn_and_tall_linear_match_torch(R, C, slope, dtype):
    """Linear + weighted BatchNorm + LeakyReLU on map rows (C ABI dva_rowbn_*) against the plain torch
    composition on the materialised per-view rows, forward and all gradients."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(R + C)
    counts = torch.randint(0, 5, (R,), generator=gen, dtype=torch.int32)
    n = float(counts.sum())
    x = torch.randn(R, C, generator=gen)
    W = torch.randn(C, C, generator=gen) * 0.3
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    gw = torch.randn(R, C, generator=gen)
    # reference: every row repeated counts times (the gathered views), standard batch norm over them
    idx = torch.repeat_interleave(torch.arange(R), counts.long())
    xr, Wr, gr, br = [t.clone().requires_grad_() for t in (x, W, gamma, beta)]
    yv = torch.nn.functional.linear(xr[idx], Wr)
    zv = torch.nn.functional.batch_norm(yv, None, None, gr, br, True, 0.1, 1e-5)
    ov = torch.nn.functional.leaky_relu(zv, slope)
    ref_grads = torch.autograd.grad((ov * gw[idx]).sum(), [xr, Wr, gr, br])
    # device
    xd, Wd, gd, bd = [t.to(DEV).requires_grad_() for t in (x, W, gamma, beta)]
    cd = counts.to(DEV)
    y = ops.tall_linear(xd.to(dtype), Wd.to(dtype))
    s1, s2 = ops.rowbn_stats(y, cd)
    mean = s1 / n
    var = (s2 / n - mean * mean).clamp_(min=0)
    out = ops.rowbn_act(y, cd, gd, bd, mean.float(), torch.rsqrt(var.float() + 1e-5), n, True, slope)
    # the loss over the views = sum over rows of counts * (row term)
    grads = torch.autograd.grad((out.float() * (gw.to(DEV) * cd.unsqueeze(1))).sum(), [xd, Wd, gd, bd])
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float32 else dict(rtol=6e-2, atol=6e-2)
    seen = counts > 0
    first = (counts.long().cumsum(0) - counts.long())[seen]      # first view of every row that is seen
    close(out[seen.to(DEV)], ov.detach()[first], **tol)
    for a, b in zip(grads, ref_grads):
        if dtype == torch.float32:
            close(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()))
        else:
            assert float((a.float().cpu() - b).norm() / b.norm()) < 5e-2


@pytest.mark.parametrize("dtype,C,G", [(torch.float32, 64, 4), (torch.bfloat16, 512, 4), (torch.bfloat16, 256, 8)])
def test_view_gather_attention_long_segments(dtype, C, G):
    """Segments longer than the register-resident short path (up to 300 views per point, one or few rows per
    team): the LDS-parked long path of the backward kernel against plain torch, forward and all gradients."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C + G)
    N, R = 300, 500
    sizes = torch.randint(0, 70, (N,), generator=gen)
    sizes[:3] = torch.tensor([300, 129, 64])          # beyond the LDS share of a team -> global fallback
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32)
    rows = torch.randn(R, C, generator=gen).to(dtype)
    compat = torch.randn(V, G, generator=gen)
    gw, gb = torch.randn(G, generator=gen), torch.randn(G, generator=gen)
    w = torch.randn(N, C, generator=gen)
    rd, cd = rows.to(DEV).requires_grad_(), compat.to(DEV).requires_grad_()
    gwd, gbd = gw.to(DEV).requires_grad_(), gb.to(DEV).requires_grad_()
    out, att, gate = ops.view_gather_attention(rd, row_idx.to(DEV), cd, csr.to(DEV), gwd, gbd)
    grads = torch.autograd.grad((out.float() * w.to(DEV)).sum(), [rd, cd, gwd, gbd])
    # torch reference on the materialised gather
    rr, cr = rows.float().requires_grad_(), compat.clone().requires_grad_()
    gwr, gbr = gw.clone().requires_grad_(), gb.clone().requires_grad_()
    a_ref = O.segment_softmax_csr(cr, csr, scaling=False)
    xp = O.segment_csr(rr[row_idx.long()] * O.expand_group_feat(a_ref, G, C), csr, 'sum')
    mx = O.segment_csr(cr, csr, 'max')
    xp = xp * O.expand_group_feat(torch.tanh(torch.relu(mx * gwr + gbr)), G, C)
    ref = torch.autograd.grad((xp * w).sum(), [rr, cr, gwr, gbr])
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=4e-2, atol=4e-2)
    close(out, xp, **tol)
    close(att, a_ref, rtol=1e-5, atol=1e-6)
    for a, b in zip(grads, ref):
        if dtype == torch.float32:
            close(a, b, **tol)
        else:
            assert float((a.float().cpu() - b).norm() / (b.norm() + 1e-6)) < 3e-2


@pytest.mark.parametrize("dtype,C", [(torch.float32, 24), (torch.bfloat16, 64), (torch.float32, 64)])
def test_gather_nearest_backward_plan_equals_atomics(dtype, C):
    """dva_gather_rows_sum (segmented reduction over the row plan) against dva_gather_nearest_bwd (atomics)."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C)
    B, H, W, P = 3, 9, 14, 5000
    x = torch.randn(B, C, H, W, generator=gen).to(dtype)
    images = torch.randint(0, B, (P,), generator=gen)
    pixels = torch.stack([torch.randint(0, W, (P,), generator=gen), torch.randint(0, H, (P,), generator=gen)], 1).short()
    w = torch.randn(P, C, generator=gen).to(dtype)
    packed = ops.pack_gather_index(images.to(DEV), torch.arange(P + 1, device=DEV), pixels.to(DEV))
    res = {}
    for algo in (0, 1):
        old, ops.ROWS_GRAD_ALGO = ops.ROWS_GRAD_ALGO, algo
        try:
            xd = x.to(DEV).requires_grad_()
            (res[algo],) = torch.autograd.grad((ops.gather_nearest(xd, packed).float() * w.to(DEV).float()).sum(), xd)
        finally:
            ops.ROWS_GRAD_ALGO = old
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    close(res[0], res[1], **tol)
    ref = torch.zeros(B, H, W, C).index_put_((images, pixels[:, 1].long(), pixels[:, 0].long()), w.float(), accumulate=True)
    close(res[0].permute(0, 2, 3, 1), ref, **tol)


@pytest.mark.parametrize("dtype,C", [(torch.float32, 24), (torch.bfloat16, 64), (torch.float32, 32),
                                     (torch.bfloat16, 32)])
def test_gather_bilinear_backward_plan_equals_atomics(dtype, C):
    """Bilinear gather backward against the atomic scatter and autograd of the oracle's sparse_interpolation:
    C = 24 -> weighted segmented reduction over the row plan of the 4 P corner taps (dva_gather_bilinear_taps +
    dva_row_plan + dva_gather_rows_sum); the other shapes -> the anchor plan (views grouped by the padded cell of
    their top-left tap: dva_gather_bilinear_taps_anchor + dva_anchor_rows_sum + dva_anchor_combine + dva_anchor_fixup),
    including views at the image borders and views WITHOUT the 2 x 2 tap structure (floor(q + 1) != floor(q) + 1,
    which the reference evaluates separately: they take the fix-up path)."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C + 1)
    B, H, W, P = 3, 9, 14, 4000
    x = torch.randn(B, C, H, W, generator=gen).to(dtype)
    images = torch.randint(0, B, (P,), generator=gen)
    pixels = torch.zeros(P, 2, dtype=torch.int16)
    coords = torch.rand(P, 2, generator=gen)
    coords[:50] = torch.tensor([0.0, 1.0])                 # borders: replication padding taps collapse
    coords[50:60] = torch.tensor([1.0, 0.0])
    # q = 9 c + 0.5 = 1 - 2^-24: floor(q) = 0 but floor(q + 1) = 2 (q + 1 rounds up to 2.0)
    coords[60:64, 0] = 0.0555555485188961
    q = coords[60, 0] * 9 + 0.5
    assert torch.floor(q + 1) != torch.floor(q) + 1
    w = torch.randn(P, C, generator=gen).to(dtype)
    packed = ops.pack_gather_index(images.to(DEV), torch.arange(P + 1, device=DEV), pixels.to(DEV))
    res = {}
    for algo in (0, 1):
        old, ops.ROWS_GRAD_ALGO = ops.ROWS_GRAD_ALGO, algo
        try:
            xd = x.to(DEV).requires_grad_()
            out = ops.gather_bilinear(xd, packed, coords.to(DEV))
            (res[algo],) = torch.autograd.grad((out.float() * w.to(DEV).float()).sum(), xd)
        finally:
            ops.ROWS_GRAD_ALGO = old
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    close(res[0], res[1], **tol)
    xr = x.float().requires_grad_()
    ref_out = O.sparse_interpolation(xr, coords, images)
    (g_ref,) = torch.autograd.grad((ref_out * w.float()).sum(), xr)
    close(res[0], g_ref, **tol)


@pytest.mark.parametrize("pix_dtype,ratio", [(torch.int16, 1.0), (torch.int16, 4.0), (torch.int32, 8.0),
                                              (torch.int64, 2.0)])
def test_mapping_row_index_equals_pack_then_row_index(pix_dtype, ratio):
    """dva_mapping_row_index == dva_pack_gather_index + dva_gather_row_index (the one-pass form the lazy gather uses),
    with several atoms per view and empty views."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(int(ratio) + 1)
    V, B, H, W = 5000, 6, 24, 40
    sizes = torch.randint(0, 4, (V,), generator=gen)
    atom_ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    P = int(atom_ptr[-1])
    images = torch.randint(0, B, (V,), generator=gen)
    pixels = torch.stack([torch.randint(0, int(W * ratio), (P,), generator=gen),
                          torch.randint(0, int(H * ratio), (P,), generator=gen)], 1).to(pix_dtype)
    packed = ops.pack_gather_index(images.to(DEV), atom_ptr.to(DEV), pixels.to(DEV), ratio=ratio)
    ref_idx, ref_counts, ref_plan = ops.gather_row_index(packed, B, H, W, with_plan=True)
    row_idx, counts, plan = ops.mapping_row_index(images.to(DEV), atom_ptr.to(DEV), pixels.to(DEV), ratio, B, H, W)
    assert torch.equal(row_idx, ref_idx) and torch.equal(counts, ref_counts)
    assert torch.equal(plan[0], ref_plan[0]) and torch.equal(plan[1], ref_plan[1])
    img_of_atom = images.repeat_interleave(sizes)
    expect = (img_of_atom * H + (pixels[:, 1].long() // int(ratio))) * W + pixels[:, 0].long() // int(ratio)
    assert torch.equal(row_idx.cpu().long(), expect)


@pytest.mark.parametrize("C", [32, 64, 128])
@pytest.mark.parametrize("gating,scaling", [(True, True), (False, False)])
def test_lean_attention_backward_fp32_equals_team_kernel(C, gating, scaling):
    """fp32 rows, four score groups: the tile-based attention backward (chain_bwd.hip attn_bwd_kernel<float>) against
    the team kernel on ragged points incl. empty ones, a 90-view point and a 400-view point (fragments), ties in the
    maximal score; and both against the PyTorch oracle."""
    from deepviewagg_amd import ops
    from oracle import pooling_oracle as O
    gen = torch.Generator().manual_seed(100 + C)
    N, G, R = 700, 4, 509
    sizes = torch.randint(0, 12, (N,), generator=gen)
    sizes[10], sizes[300] = 90, 400
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    rows = torch.randn(R, C, generator=gen)
    row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32)
    compat = torch.randn(V, G, generator=gen)
    compat[csr[20]:csr[20] + 2] = compat[csr[20]]                 # a tie: the first maximal view takes the gate path
    w = torch.randn(N, C, generator=gen)
    gw = torch.randn(1, G, generator=gen) if gating else None
    gb = torch.randn(1, G, generator=gen) if gating else None

    def run(lean):
        ops.LEAN_ATTENTION_BWD = lean
        try:
            r = rows.to(DEV).requires_grad_()
            c = compat.to(DEV).requires_grad_()
            pw = gw.to(DEV).requires_grad_() if gating else None
            pb = gb.to(DEV).requires_grad_() if gating else None
            out, att, gate = ops.view_gather_attention(r, row_idx.to(DEV), c, csr.to(DEV), pw, pb, scaling=scaling)
            grads = torch.autograd.grad((out * w.to(DEV)).sum(), [r, c] + ([pw, pb] if gating else []))
            return [out.detach()] + [g_.detach() for g_ in grads]
        finally:
            ops.LEAN_ATTENTION_BWD = True
    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        torch.testing.assert_close(x, y, rtol=2e-4, atol=2e-5)
    # oracle
    gate_m = O.Gating(G) if gating else None
    if gating:
        with torch.no_grad():
            gate_m.weight.copy_(gw)
            gate_m.bias.copy_(gb)
    r0, c0 = rows.clone().requires_grad_(), compat.clone().requires_grad_()
    out_ref, _, _ = O.attention_tail(r0[row_idx.long()], c0, csr, gate_m, G, C, scaling)
    g_ref = torch.autograd.grad((out_ref * w).sum(), [r0, c0] + (list(gate_m.parameters()) if gating else []))
    torch.testing.assert_close(a[0].cpu(), out_ref.detach(), rtol=1e-4, atol=1e-5)
    for x, y in zip(a[1:], g_ref):
        torch.testing.assert_close(x.cpu().reshape(y.shape), y, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("C,G", [(32, 4), (64, 4), (64, 2), (128, 1), (512, 4)])
@pytest.mark.parametrize("gating,scaling", [(True, True), (False, False)])
def test_lean_attention_backward_bf16_equals_team_kernel(C, G, gating, scaling):
    """bf16 rows, 1 / 2 / 4 score groups (round 4: the attention backward of QKVBimodalCSRPool on the chain): the chain
    path's tile-based attention backward + 16-byte-record rows gradient against the team kernels (same bf16 inputs: the
    differences are the bf16 gate x attention weights of the records and summation orders) and against the fp32 PyTorch
    oracle evaluated on the same bf16-rounded rows."""
    from deepviewagg_amd import ops
    from oracle import pooling_oracle as O
    gen = torch.Generator().manual_seed(200 + C + G)
    N, R = 700, 509
    sizes = torch.randint(0, 12, (N,), generator=gen)
    sizes[10], sizes[300] = 90, 400
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    rows = torch.randn(R, C, generator=gen).bfloat16()
    row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32)
    compat = torch.randn(V, G, generator=gen)
    compat[csr[20]:csr[20] + 2] = compat[csr[20]]
    w = torch.randn(N, C, generator=gen).bfloat16()
    gw = torch.randn(1, G, generator=gen) if gating else None
    gb = torch.randn(1, G, generator=gen) if gating else None

    def run(lean):
        ops.LEAN_ATTENTION_BWD = lean
        try:
            r = rows.to(DEV).requires_grad_()
            c = compat.to(DEV).requires_grad_()
            pw = gw.to(DEV).requires_grad_() if gating else None
            pb = gb.to(DEV).requires_grad_() if gating else None
            out, att, gate = ops.view_gather_attention(r, row_idx.to(DEV), c, csr.to(DEV), pw, pb, scaling=scaling)
            assert out.dtype == torch.bfloat16
            grads = torch.autograd.grad((out.float() * w.to(DEV).float()).sum(), [r, c] + ([pw, pb] if gating else []))
            return [out.detach()] + [g_.detach() for g_ in grads]
        finally:
            ops.LEAN_ATTENTION_BWD = True
    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0])
    rel = lambda x, y: float((x.float() - y.float()).norm() / y.float().norm().clamp_min(1e-20))
    assert rel(a[1], b[1]) < 6e-3, rel(a[1], b[1])                   # rows gradient: bf16 weights in the records
    for x, y in zip(a[2:], b[2:]):
        assert rel(x, y) < 2e-3, rel(x, y)
    # oracle (fp32 on the bf16-rounded rows; the pooled output is rounded to bf16 by the kernels)
    gate_m = O.Gating(G) if gating else None
    if gating:
        with torch.no_grad():
            gate_m.weight.copy_(gw)
            gate_m.bias.copy_(gb)
    r0, c0 = rows.float().clone().requires_grad_(), compat.clone().requires_grad_()
    out_ref, _, _ = O.attention_tail(r0[row_idx.long()], c0, csr, gate_m, G, C, scaling)
    g_ref = torch.autograd.grad((out_ref * w.float()).sum(), [r0, c0] + (list(gate_m.parameters()) if gating else []))
    assert rel(a[0].cpu(), out_ref.detach()) < 4e-3
    assert rel(a[1].cpu(), g_ref[0]) < 8e-3
    for x, y in zip(a[2:], g_ref[1:]):
        assert rel(x.cpu().reshape(y.shape), y) < 4e-3, rel(x.cpu().reshape(y.shape), y)


@pytest.mark.parametrize("C", [32, 64])
def test_bilinear_scatter_with_batchnorm_backward(C):
    """ops.bilinear_scatter(bn_backward=...): the BatchNorm_a backward of the fused bilinear path folded into the anchor
    scatter, row by row from the stored z_a and at the level of the anchor (4 x 4 Gram matrix of the tap weights x the
    four rows of Y), against index_add of the explicitly transformed rows -- including border views and views without
    the 2 x 2 tap structure (dummy anchor: fix-up kernel)."""
    from deepviewagg_amd import ops, fused_bilinear
    gen = torch.Generator().manual_seed(C)
    B, H, W, P = 3, 9, 14, 5000
    x = torch.randn(B, C, H, W, generator=gen).bfloat16()
    images = torch.randint(0, B, (P,), generator=gen)
    pixels = torch.zeros(P, 2, dtype=torch.int16)
    coords = torch.rand(P, 2, generator=gen)
    coords[:50] = torch.tensor([0.0, 1.0])
    coords[50:60] = torch.tensor([1.0, 0.0])
    coords[60:64, 0] = 0.0555555485188961           # floor(q + 1) != floor(q) + 1: no 2 x 2 structure
    packed = ops.pack_gather_index(images.to(DEV), torch.arange(P + 1, device=DEV), pixels.to(DEV))
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    lazy = ops.lazy_gather_bilinear(xd, packed, coords.to(DEV), exact=True)
    rows4, w4, anchors = lazy.tap_rows, lazy.tap_weights, lazy.anchors
    assert int((anchors == B * (H + 1) * (W + 1)).sum()) >= 4, "the case must contain dummy-anchor views"
    Y = lazy.rows                                                            # bf16 [R, C]: stands for Linear_a(x)
    z_full = (Y.float()[rows4.long()] * w4.unsqueeze(-1)).sum(1)            # [P, C] fp32
    z_a = z_full.bfloat16()
    dy = torch.randn(P, C, generator=gen).bfloat16().to(DEV)
    bn = torch.stack([torch.randn(C, generator=gen) * 0.2, torch.rand(C, generator=gen) + 0.5,
                      torch.randn(C, generator=gen) * 0.3 + 1.0, torch.randn(C, generator=gen)]).to(DEV)
    sm = (torch.randn(2 * C, generator=gen) * 0.05).to(DEV)
    kappa = fused_bilinear.position_order(C, DEV)                            # channel held by position p
    mean, inv, gam = bn[0][kappa], bn[1][kappa], bn[2][kappa]
    s1, s2 = sm[:C][kappa], sm[C:][kappa]
    g = gam * inv
    k1, k2 = g * (s1 - mean * inv * s2), g * inv * s2

    def reference(z):
        dz = g * dy.float() - k1 - k2 * z
        out = torch.zeros(B * H * W, C, device=DEV)
        for k in range(4):
            out.index_add_(0, rows4[:, k].long(), dz * w4[:, k:k + 1])
        return out
    got_rows = ops.bilinear_scatter(dy, rows4, w4, anchors, B, H, W, bn_backward=(z_a, bn, sm, None))
    got_gram = ops.bilinear_scatter(dy, rows4, w4, anchors, B, H, W, bn_backward=(z_a, bn, sm, Y))
    ref_rows = reference(z_a.float())
    close(got_rows, ref_rows, rtol=1e-4, atol=1e-4)
    # the anchor-level form sees the unrounded z_a on the views with the 2 x 2 structure
    ref_gram = reference(torch.where((anchors == B * (H + 1) * (W + 1)).unsqueeze(1), z_a.float(), z_full))
    close(got_gram, ref_gram, rtol=1e-4, atol=2e-4)
    assert float((got_gram - got_rows).norm() / got_rows.norm()) < 5e-3


@pytest.mark.parametrize("dtype,C", [(torch.bfloat16, 64), (torch.bfloat16, 256), (torch.float32, 32), (torch.bfloat16, 20),
                                     (torch.float32, 24), (torch.float32, 6)])
def test_gather_bilinear_forward_is_bitwise_the_reference_expression(dtype, C):
    """dva_gather_bilinear_fwd, vectorised (C / VEC a power of two: 16-byte accesses, taps once per lane) and scalar
    kernels: out = ((w_tl X_tl + w_tr X_tr) + w_bl X_bl) + w_br X_br evaluated in fp32 in this order without fma, rounded
    once to the feature dtype -- bit-identical to the same expression in torch on the taps of dva_gather_bilinear_taps."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C)
    B, H, W, P = 3, 9, 14, 7001
    x = torch.randn(B, C, H, W, generator=gen).to(dtype)
    images = torch.randint(0, B, (P,), generator=gen)
    pixels = torch.zeros(P, 2, dtype=torch.int16)
    coords = torch.rand(P, 2, generator=gen)
    coords[:50] = torch.tensor([0.0, 1.0])
    coords[50:60] = torch.tensor([1.0, 0.0])
    packed = ops.pack_gather_index(images.to(DEV), torch.arange(P + 1, device=DEV), pixels.to(DEV))
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    out = ops.gather_bilinear(xd, packed, coords.to(DEV))
    lazy = ops.lazy_gather_bilinear(xd, packed, coords.to(DEV), exact=True)
    rows = lazy.rows.float()
    r4, w4 = lazy.tap_rows.long(), lazy.tap_weights
    acc = w4[:, 0:1] * rows[r4[:, 0]]
    acc = acc + w4[:, 1:2] * rows[r4[:, 1]]
    acc = acc + w4[:, 2:3] * rows[r4[:, 2]]
    acc = acc + w4[:, 3:4] * rows[r4[:, 3]]
    assert torch.equal(out, acc.to(dtype))


@pytest.mark.parametrize("V,R,C,G,how", [(9000, 600, 64, 4, "uniform"), (300000, 5000, 32, 2, "uniform"),
                                         (2000000, 1 << 18, 64, 4, "uniform"), (500000, 70000, 64, 1, "skewed"),
                                         (100000, 2000, 32, 4, "one_row")])
def test_bucket_rows_grad_f32_equals_index_add(V, R, C, G, how):
    """The fp32 twin of the bucket rows gradient (round 6: dva_plan_split_sort_records32 + dva_plan_split_rows_grad with
    fp32 / fp32): the 32-byte records of the fp32 attention backward {point | 4 fp32 weights | pad} against an fp64
    ``index_add`` of the same products (the reference's backward of the row gather, core/multimodal/image.py:1262-1287),
    and against the segmented reduction over the permutation plan (dva_view_gather_rows_grad) it replaces."""
    from deepviewagg_amd import ops, _lib
    from deepviewagg_amd._lib import ptr, check
    gen = torch.Generator().manual_seed(V + R + C + 1)
    N = max(V // 8, 4)
    if how == "uniform":
        row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32)
    elif how == "one_row":
        row_idx = torch.full((V,), R - 3, dtype=torch.int32)
    else:
        hot = torch.randint(0, R, (16,), generator=gen)
        row_idx = torch.where(torch.rand(V, generator=gen) < 0.5, hot[torch.randint(0, 16, (V,), generator=gen)],
                              torch.randint(0, R, (V,), generator=gen)).to(torch.int32)
    point = torch.randint(0, N, (V,), generator=gen, dtype=torch.int32)
    wts = torch.randn(V, 4, generator=gen)
    rec = torch.zeros(V, 8, dtype=torch.float32)
    rec[:, 0] = point.view(torch.float32)
    rec[:, 1:5] = wts
    rec[:, 5:] = float("nan")                              # the pad words must not matter
    gout = torch.randn(N, C, generator=gen).to(DEV)
    rd, recd = row_idx.to(DEV), rec.to(DEV)
    old = ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS
    try:
        ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS = True, 0
        plan = ops.row_plan(rd, R, with_counts=False)[0]
        assert isinstance(plan, ops.SplitPlan)
        st = torch.cuda.current_stream().cuda_stream
        a = plan.rows_grad_f32(gout, recd.clone(), C, G, st)
        a2 = plan.rows_grad_f32(gout, recd.clone(), C, G, st)
    finally:
        ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS = old
    assert a is not None and a.dtype == torch.float32 and a.shape == (R, C)
    assert torch.equal(a, a2)                                       # deterministic
    ch_group = torch.arange(C) // (C // G)
    ref = torch.zeros(R, C, dtype=torch.float64)
    for lo in range(0, V, 500000):
        sl = slice(lo, lo + 500000)
        ref.index_add_(0, row_idx[sl].long(), gout.cpu().double()[point[sl].long()] * wts[sl].double()[:, ch_group])
    scale = float(ref.abs().max()) + 1e-9
    assert float((a.cpu().double() - ref).abs().max()) <= scale * 2e-5        # fp32 sums of up to ~10^5 terms (one_row)
    # the permutation-plan kernel on the same records
    (perm, row_ptr), _ = ops.row_plan(rd, R, with_counts=False, split=False)
    b = torch.empty((R, C), dtype=torch.float32, device=DEV)
    lib = _lib.load()
    check(lib.dva_view_gather_rows_grad(ptr(gout), None, None, None, ptr(perm), ptr(row_ptr), ptr(recd), 8, ptr(b), R, V, C,
                                        G, _lib.DVA_F32, st), "dva_view_gather_rows_grad")
    assert float((a - b).abs().max()) <= scale * 2e-5


@pytest.mark.parametrize("C,gating", [(64, True), (32, False)])
def test_view_gather_attention_f32_split_plan_equals_permutation_plan(C, gating):
    """ops.view_gather_attention on fp32 rows (the lean fp32 backward, G = 4) over the split plan (32-byte records through
    pass A + the fp32 bucket kernel) = over the permutation plan, to fp32 rounding."""
    from deepviewagg_amd import ops
    gen = torch.Generator().manual_seed(C + 7)
    N, R, G = 6000, 700, 4
    sizes = torch.randint(0, 9, (N,), generator=gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(DEV)
    V = int(csr[-1])
    row_idx = torch.randint(0, R - 20, (V,), generator=gen, dtype=torch.int32).to(DEV)
    rows = torch.randn(R, C, generator=gen)
    compat = torch.randn(V, G, generator=gen)
    gw = torch.randn(G, generator=gen) if gating else None
    gb = torch.randn(G, generator=gen) if gating else None
    w = torch.randn(N, C, generator=gen).to(DEV)

    def run(split):
        old = ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS
        ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS = split, 0
        calls = {"f32": 0}
        orig = ops.SplitPlan.rows_grad_f32

        def counted(self, *a, **k):
            r = orig(self, *a, **k)
            calls["f32"] += r is not None
            return r
        ops.SplitPlan.rows_grad_f32 = counted
        try:
            plan = ops.row_plan(row_idx, R, with_counts=False, split=split)[0]
            assert isinstance(plan, ops.SplitPlan) == split
            rd = rows.to(DEV).requires_grad_()
            cd = compat.to(DEV).requires_grad_()
            gwd = gw.to(DEV).requires_grad_() if gating else None
            gbd = gb.to(DEV).requires_grad_() if gating else None
            out, _, _ = ops.view_gather_attention(rd, row_idx, cd, csr, gwd, gbd, plan=plan)
            res = [out] + list(torch.autograd.grad((out * w).sum(), [rd, cd]))
        finally:
            ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS = old
            ops.SplitPlan.rows_grad_f32 = orig
        assert calls["f32"] == (1 if split else 0)
        return res

    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    torch.testing.assert_close(b[1], a[1], rtol=1e-5, atol=1e-5 * float(a[1].abs().max()))
    assert float(b[1][R - 20:].abs().max()) == 0.0 and float(b[1].abs().max()) > 0.0
