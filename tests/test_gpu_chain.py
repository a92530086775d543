"""-m gpu: the bf16 recompute chain (csrc/chain_*.hip through fused_chain.py) against the CPU oracle.

Tolerances: the chain feeds bf16-rounded operands to the matrix cores (activations AND weights, like the reference's
Linear layers under torch.autocast(bfloat16)) and keeps accumulation / BatchNorm / softmax in fp32; the value rows
are bf16.  Outputs are compared with the fp32 oracle evaluated on the same bf16-rounded value rows: relative L2
error <= 2e-2 (measured ~3e-3), and against the oracle's own autocast error where stated."""
import numpy as np
import pytest
import torch

from oracle import pooling_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def make_case(seed, N, C, sizes_fn, B=3, H=12, W=20):
    gen = torch.Generator().manual_seed(seed)
    sizes = sizes_fn(N, gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    images = torch.randint(0, B, (V,), generator=gen)
    pixels = torch.stack([torch.randint(0, W, (V,), generator=gen), torch.randint(0, H, (V,), generator=gen)], 1).short()
    x = torch.randn(B, C, H, W, generator=gen).bfloat16().float()
    x_map = torch.rand(V, 8, generator=gen)
    w = torch.randn(N, C, generator=gen)
    return dict(gen=gen, csr=csr, V=V, images=images, pixels=pixels, x=x, x_map=x_map, w=w, N=N, C=C)


def ragged(N, gen):
    return torch.randint(0, 9, (N,), generator=gen)


def ragged_long(N, gen):
    s = torch.randint(0, 7, (N,), generator=gen)
    s[5] = 100
    s[6] = 33
    s[7] = 64
    s[N - 1] = 70
    return s


def full32(N, gen):
    return torch.full((N,), 32, dtype=torch.long)


def build(case, G, train, gating=True, scaling=True, seed=5):
    from deepviewagg_amd.modules.multimodal import pooling as P
    gen = torch.Generator().manual_seed(seed)
    kwargs = dict(in_map=8, in_mod=case["C"], num_groups=G, use_num=True, gating=gating, group_scaling=scaling)
    ref = O.GroupBimodalCSRPool(**kwargs)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.4)
        for n, b in ref.named_buffers():
            if "running_mean" in n:
                b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
            if "running_var" in n:
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)
    ref.train(train)
    m = P.GroupBimodalCSRPool(**kwargs)
    m.load_state_dict(ref.state_dict(), strict=True)
    return ref, m.to(DEV).train(train)


def run_dev(case, m, chain, need_grad=True):
    from deepviewagg_amd import ops, fused_chain
    from deepviewagg_amd.modules.multimodal import pooling as P
    V = case["V"]
    xd = case["x"].to(DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(need_grad)
    packed = ops.pack_gather_index(case["images"].to(DEV), torch.arange(V + 1, device=DEV), case["pixels"].to(DEV))
    fused_chain.FORCE = chain
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lazy = ops.lazy_gather_nearest(xd, packed, exact=True)
            lazy = P.BimodalCSRPool(mode='max')(None, lazy, None, torch.arange(V + 1, device=DEV))
            out = m(None, lazy, case["x_map"].to(DEV), case["csr"].to(DEV))
        grads = None
        if need_grad:
            grads = torch.autograd.grad((out.float() * case["w"].to(DEV)).sum(), [xd] + list(m.parameters()),
                                        allow_unused=True)
    finally:
        fused_chain.FORCE = None
    return out, grads


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("sizes_fn,N,C,G", [(ragged, 3000, 64, 4), (ragged_long, 2000, 64, 4), (full32, 300, 64, 4),
                                            (ragged_long, 1500, 32, 2), (ragged, 1500, 128, 1),
                                            (ragged_long, 700, 512, 4), (ragged, 900, 256, 4)])
def test_chain_forward_matches_oracle(sizes_fn, N, C, G, train):
    case = make_case(3, N, C, sizes_fn)
    ref, m = build(case, G, train)
    out_ref = ref(None, O.gather_nearest(case["x"], case["images"], case["pixels"]), case["x_map"], case["csr"])
    out, _ = run_dev(case, m, chain=True, need_grad=False)
    assert out.dtype == torch.bfloat16
    r = rel(out, out_ref)
    ref2, _ = build(case, G, train)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out_amp = ref2(None, O.gather_nearest(case["x"], case["images"], case["pixels"]), case["x_map"], case["csr"])
    r_amp = rel(out_amp, out_ref)
    print(f"chain fwd rel err {r:.4f}  (reference under autocast: {r_amp:.4f})")
    assert r < max(2e-2, 1.5 * r_amp), (r, r_amp)
    # unseen points stay zero
    unseen = (case["csr"][1:] == case["csr"][:-1])
    assert float(out.float().cpu()[unseen].abs().max() if unseen.any() else 0.0) == 0.0
    if train:
        for (k, a), b in zip(m.state_dict().items(), ref.state_dict().values()):
            if "running" in k:
                torch.testing.assert_close(a.cpu(), b, rtol=2e-2, atol=2e-3)


def test_tile_table_properties():
    from deepviewagg_amd import fused_chain
    gen = torch.Generator().manual_seed(0)
    for sizes in (ragged_long(5000, gen), full32(257, gen), torch.zeros(100, dtype=torch.long),
                  torch.randint(0, 3, (100000,), generator=gen), torch.randint(20, 45, (3000,), generator=gen)):
        csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(DEV)
        V = int(csr[-1])
        tiles, n = fused_chain.build_tiles(csr, V)
        T = int(n)
        t = tiles[:T].cpu().numpy()
        v0, nv, frag = t[:, 0], t[:, 1] & 0xff, t[:, 1] >> 8
        assert (nv >= 1).all() and (nv <= 32).all()
        assert T == 0 or v0[0] == 0
        assert (v0[1:] == v0[:-1] + nv[:-1]).all() and (T == 0 or v0[-1] + nv[-1] == V)   # exact cover, in order
        starts = set(csr.cpu().numpy().tolist())
        for i in range(T):
            if frag[i] == 0:
                assert v0[i] in starts and (v0[i] + nv[i]) in starts
            if frag[i] in (1, 2):
                assert nv[i] == 32 and frag[i + 1] in (2, 3)
            if frag[i] == 1:
                assert v0[i] in starts
            if frag[i] == 3:
                assert (v0[i] + nv[i]) in starts
        # no point of <= 32 views is split: no tile boundary strictly inside it
        ptr = csr.cpu().numpy()
        small = ((ptr[1:] - ptr[:-1]) <= 32) & (ptr[1:] > ptr[:-1])
        lo = np.searchsorted(v0, ptr[:-1], side='right')     # first cut > start
        hi = np.searchsorted(v0, ptr[1:], side='left')       # first cut >= end
        assert (lo[small] == hi[small]).all()
