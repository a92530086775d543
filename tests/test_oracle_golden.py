"""The CPU oracle (oracle/pooling_oracle.py) pinned against the reference's own outputs.

tests/golden/*.npz were produced by oracle/gen_golden.py, which imports and runs the reference's
Python source (pooling.py, modules.py, image.py ...) in the build container.  Here the oracle's
restatement must reproduce those outputs and gradients on the same inputs and weights.
"""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden, t, state_dict_from
from oracle import pooling_oracle as O

TOL = dict(rtol=1e-5, atol=1e-6)


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    torch.testing.assert_close(a, t(b) if isinstance(b, np.ndarray) else b, **tol)


def test_softmax_known_answers():
    """pooling.py:913-921 docstring example; values quoted in SURVEY.md §4."""
    g = load_golden("softmax_known")
    src, csr = t(g["src"]), t(g["csr"])
    out = O.segment_softmax_csr(src, csr)
    close(out, g["out"])
    close(out[:5, 0], torch.tensor([0.011656, 0.031685, 0.086129, 0.234122, 0.636409]), atol=1e-6, rtol=1e-4)
    out_s = O.segment_softmax_csr(src, csr, scaling=True)
    close(out_s, g["out_scaled"])
    close(out_s[:5, 0], torch.tensor([0.067486, 0.105545, 0.165067, 0.258157, 0.403744]), atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("G", [1, 4])
def test_softmax_random(G):
    g = load_golden(f"softmax_random_G{G}")
    csr, w = t(g["csr"]), t(g["w"])
    for sc in (0, 1):
        src = t(g["src"]).requires_grad_()
        out = O.segment_softmax_csr(src, csr, scaling=bool(sc))
        close(out, g[f"out_{sc}"])
        (gr,) = torch.autograd.grad((out * w).sum(), src)
        close(gr, g[f"grad_{sc}"])


def test_segment_csr_and_gather():
    g = load_golden("segment_csr")
    csr = t(g["csr"])
    for red in ("sum", "mean", "max", "min"):
        src = t(g["src"]).requires_grad_()
        out = O.segment_csr(src, csr, red)
        close(out, g[f"out_{red}"])
        (gr,) = torch.autograd.grad((out * t(g[f"w_{red}"])).sum(), src)
        close(gr, g[f"grad_{red}"])
    close(O.gather_csr(t(g["gather_src"]), csr), g["gather_out"])
    # the loop definition and the vectorised arg agree (ties -> first row)
    for red in ("max", "min"):
        assert torch.equal(O.segment_arg(t(g["src"]), csr, red), O.segment_arg_fast(t(g["src"]), csr, red))


POOL_CASES = ["pool_group_default_train", "pool_group_default_eval", "pool_group_docstring",
              "pool_group_usemod_nogate", "pool_group_mlpset_g1", "pool_group_minmaxpool",
              "pool_qkv_default", "pool_qkv_modqk"]


def build_oracle_pool(name, g):
    kwargs = ast.literal_eval(str(g["kwargs"]))
    cls = O.QKVBimodalCSRPool if "qkv" in name else O.GroupBimodalCSRPool
    m = cls(**kwargs)
    m.load_state_dict(state_dict_from(g), strict=True)
    m.train(bool(g["train"]))
    return m, kwargs


@pytest.mark.parametrize("name", POOL_CASES)
def test_pool_modules(name):
    g = load_golden(name)
    m, _ = build_oracle_pool(name, g)
    csr = t(g["csr"])
    x_mod, x_map = t(g["x_mod"]).requires_grad_(), t(g["x_map"]).requires_grad_()
    x_main = t(g["x_main"]).requires_grad_() if "x_main" in g else None
    out = m(x_main, x_mod, x_map, csr)
    close(out, g["out"], rtol=1e-4, atol=1e-5)
    close(m.last_C, g["last_C"], rtol=1e-4, atol=1e-5)
    close(m.last_A, g["last_A"], rtol=1e-4, atol=1e-5)
    if m.G is not None:
        close(m.last_G, g["last_G"], rtol=1e-4, atol=1e-5)
    ins = [x_mod, x_map] + ([x_main] if x_main is not None else [])
    names = [n for n, _ in m.named_parameters()]
    grads = torch.autograd.grad((out * t(g["w"])).sum(), ins + list(m.parameters()), allow_unused=True)
    close(grads[0], g["grad_x_mod"], rtol=1e-3, atol=1e-5)
    close(grads[1], g["grad_x_map"], rtol=1e-3, atol=1e-5)
    if x_main is not None:
        close(grads[2], g["grad_x_main"], rtol=1e-3, atol=1e-5)
    for n, gr in zip(names, grads[len(ins):]):
        ref = t(g["gp/" + n])
        gr = gr if gr is not None else torch.zeros_like(ref)
        close(gr, ref, rtol=2e-3, atol=2e-5)
    # BatchNorm running statistics after the forward
    for k, v in m.state_dict().items():
        if "running" in k:
            close(v, g["sd_after/" + k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["pool_group_c64_train", "pool_group_c64_eval"])
def test_pool_headline_shape(name):
    """The headline instantiation (C = 64, G = 4, points with exactly 32 and with more than 32 views, unseen points):
    the oracle against the reference's own forward + backward (oracle/gen_golden.py pools_headline)."""
    g = load_golden(name)
    m, _ = build_oracle_pool(name, g)
    csr = t(g["csr"])
    x_mod = t(g["x_mod"]).requires_grad_()
    out = m(None, x_mod, t(g["x_map"]), csr)
    close(out, g["out"], rtol=1e-4, atol=1e-5)
    names = [n for n, _ in m.named_parameters()]
    grads = torch.autograd.grad((out * t(g["w"])).sum(), [x_mod] + list(m.parameters()), allow_unused=True)
    close(grads[0], g["grad_x_mod"], rtol=1e-3, atol=1e-5)
    for n, gr in zip(names, grads[1:]):
        ref = t(g["gp/" + n])
        gr = gr if gr is not None else torch.zeros_like(ref)
        close(gr, ref, rtol=2e-3, atol=5e-5)
    for k, v in m.state_dict().items():
        if "running" in k:
            close(v, g["sd_after/" + k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["pool_group_bilinear_train", "pool_group_bilinear_eval"])
def test_pool_bilinear_shape(name):
    """The fused bilinear path's shape (sparse_interpolation -> E_mod 128 -> 32 per view -> view attention; points with
    32 / 40 / 70 views, border pixels): the oracle against the reference's own forward + backward
    (oracle/gen_golden.py pools_bilinear)."""
    g = load_golden(name)
    m, _ = build_oracle_pool(name, g)
    csr = t(g["csr"])
    x = t(g["x"]).requires_grad_()
    x_mod = O.gather_bilinear(x, t(g["images"]), t(g["pixels"]), tuple(int(v) for v in g["mapping_size"]))
    close(x_mod[:64], g["x_interp_head"], rtol=1e-5, atol=1e-6)
    out = m(None, x_mod, t(g["x_map"]), csr)
    close(out, g["out"], rtol=1e-4, atol=1e-5)
    names = [n for n, _ in m.named_parameters()]
    grads = torch.autograd.grad((out * t(g["w"])).sum(), [x] + list(m.parameters()), allow_unused=True)
    close(grads[0], g["grad_x"], rtol=1e-3, atol=1e-5)
    for n, gr in zip(names, grads[1:]):
        ref = t(g["gp/" + n])
        gr = gr if gr is not None else torch.zeros_like(ref)
        close(gr, ref, rtol=2e-3, atol=5e-5)
    for k, v in m.state_dict().items():
        if "running" in k:
            close(v, g["sd_after/" + k], rtol=1e-4, atol=1e-6)


def test_simple_pools_and_fusion():
    g = load_golden("pool_simple")
    csr, x_mod, x_map = t(g["csr"]), t(g["x_mod"]), t(g["x_map"])
    for mode in ("max", "mean", "min", "sum"):
        close(O.bimodal_csr_pool(x_mod, csr, mode), g[f"pool_{mode}"])
    for mode in ("max", "min"):
        close(O.heuristic_csr_pool(x_mod, x_map, csr, mode, 0), g[f"heur_{mode}_0"])
        close(O.heuristic_csr_pool(x_mod, x_map, csr, mode, 7), g[f"heur_{mode}_occlusion"])
    a, b = t(g["fusion_a"]), t(g["fusion_b"])
    for mode in ("residual", "concatenation", "both", "modality"):
        close(O.bimodal_fusion(a, b, mode), g[f"fusion_{mode}"])


def images_per_atom(g):
    sizes = t(g["atom_pointers"])[1:] - t(g["atom_pointers"])[:-1]
    return t(g["images"]).repeat_interleave(sizes)


def test_gather_nearest_and_bilinear():
    g = load_golden("gather")
    ipa, pix = images_per_atom(g), t(g["pixels"])
    x = t(g["x"]).requires_grad_()
    out = O.gather_nearest(x, ipa, pix, float(g["downscale"]))
    assert torch.equal(out, t(g["out_nearest"]))
    (gr,) = torch.autograd.grad((out * t(g["w_nearest"])).sum(), x)
    close(gr, g["grad_x_nearest"])
    out = O.gather_bilinear(x, ipa, pix, tuple(g["mapping_size"].tolist()))
    close(out, g["out_bilinear"], rtol=1e-6, atol=1e-6)
    (gr,) = torch.autograd.grad((out * t(g["w_bilinear"])).sum(), x)
    close(gr, g["grad_x_bilinear"])


def test_gather_multipixel_atomic_max():
    g = load_golden("gather_multipixel")
    out = O.gather_nearest(t(g["x"]), images_per_atom(g), t(g["pixels"]), float(g["downscale"]))
    assert torch.equal(out, t(g["out_nearest"]))
    close(O.segment_csr(out, t(g["atom_pointers"]), "max"), g["out_atomic_max"])


@pytest.mark.parametrize("tag", ["surf", "dup"])
def test_knn_oracle_matches_reference_neighborhood_features(tag):
    """oracle/knn_oracle.py against the reference's own NeighborhoodBasedMappingFeatures._process
    (core/data_transform/multimodal/image.py:482-612; fixture written by oracle/gen_golden.py neighborhood through the
    KeOps branch with a brute-force argKmin): pins density + occlusion (mapping features 7-8)."""
    from oracle import knn_oracle as KO
    g = load_golden("neighborhood")
    k_list = [int(k) for k in g["k_list"]]
    xyz = g[f"{tag}_xyz"]
    nbr, _ = KO.knn_bruteforce(xyz, k_list[-1])
    got = KO.neighborhood_features(xyz, g[f"{tag}_pointers"], g[f"{tag}_images"], nbr, k_list, voxel=float(g["voxel"]))
    ref = t(g[f"{tag}_features_out"])[:, 3:]
    torch.testing.assert_close(got[:, :2], ref[:, :2], rtol=1e-6, atol=0)        # densities (inf for zero radii: equal)
    assert torch.equal(got[:, 2:], ref[:, 2:])                                    # occlusions: counts / (k + 1)
