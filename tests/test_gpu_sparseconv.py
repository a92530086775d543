"""Sparse 3D convolution on the GPU (C ABI dva_voxel_kernel_map / dva_sparse_conv_apply / dva_sparse_conv_wgrad,
modules/SparseConv3d) against oracle/sparseconv_oracle.py (torchsparse 1.1.0 is not in the reference tree:
parity unpinned for the library itself; the oracle is pinned to torch's dense conv3d in
tests/test_sparseconv_oracle.py).  Kernel maps: bit-exact.  Features: fp32 path within 2e-5 of the fp64
oracle relative to the output scale (3-term bf16 split on the matrix cores), bf16 path within bf16 rounding
of the oracle evaluated on the same bf16-rounded operands."""
import copy

import numpy as np
import pytest
import torch

from oracle import sparseconv_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def surface_cloud(n, extent, seed, batches=1, stride=1, lo=0):
    """Voxels near the faces of a box (the occupancy pattern of scanned rooms), unique rows (x, y, z, b)."""
    rng = np.random.default_rng(seed)
    p = rng.integers(lo, lo + extent, size=(n, 3))
    face = rng.integers(0, 3, n)
    p[np.arange(n), face] = lo + rng.integers(0, 2, n) * (extent - 1)
    b = rng.integers(0, batches, size=(n, 1))
    c = np.unique(np.concatenate([p * stride, b], 1), axis=0)
    rng.shuffle(c)
    return torch.from_numpy(c.astype(np.int32))


@pytest.mark.parametrize("n,extent,k,stride,batches,lo", [(1, 4, 3, 1, 1, 0), (700, 10, 3, 1, 2, -5),
                                                          (900, 12, 2, 2, 2, -3), (500, 9, 3, 2, 1, 0)])
def test_kernel_map_matches_oracle(n, extent, k, stride, batches, lo):
    from deepviewagg_amd import ops
    from deepviewagg_amd.modules.SparseConv3d.nn import downsample_coords, kernel_offsets
    src = surface_cloud(n, extent, seed=n + k, batches=batches, lo=lo)
    dst = src if stride == 1 else downsample_coords(src.to(DEV), stride).cpu()
    assert torch.equal(dst, src if stride == 1 else O.downsample_coords(src, stride))
    offs = kernel_offsets(k, 1)
    nbr = ops.voxel_kernel_map(src.to(DEV), dst.to(DEV), offs)
    assert nbr.dtype == torch.int32 and torch.equal(nbr.cpu(), O.kernel_map(src, dst, offs))
    nbr_t = ops.voxel_kernel_map(dst.to(DEV), src.to(DEV), -offs)
    assert torch.equal(nbr_t.cpu(), O.kernel_map(dst, src, -offs))


def _maps(coords, k, stride):
    from deepviewagg_amd import ops
    from deepviewagg_amd.modules.SparseConv3d.nn import downsample_coords, kernel_offsets
    c = coords.to(DEV)
    dst = c if stride == 1 else downsample_coords(c, stride)
    offs = kernel_offsets(k, 1)
    return ops.voxel_kernel_map(c, dst, offs), ops.voxel_kernel_map(dst, c, -offs)


@pytest.mark.parametrize("n,cin,cout,k,stride,bias", [
    (50, 16, 16, 3, 1, False), (1300, 32, 64, 3, 1, True), (777, 5, 7, 3, 1, True), (900, 64, 128, 3, 1, False),
    (600, 80, 48, 3, 1, False), (1500, 32, 32, 2, 2, False), (400, 144, 16, 2, 2, True)])
def test_sparse_conv_fp32_matches_oracle(n, cin, cout, k, stride, bias):
    from deepviewagg_amd import ops
    torch.manual_seed(n + cin)
    coords = surface_cloud(n, 14, seed=n, batches=2)
    nbr, nbr_t = _maps(coords, k, stride)
    x = torch.randn(coords.shape[0], cin)
    W = torch.randn(k ** 3, cin, cout) / np.sqrt(cin * k ** 3 / 4)
    b = torch.randn(cout) if bias else None
    g = torch.randn(nbr.shape[1], cout)
    # oracle in float64
    xr, Wr = x.double().requires_grad_(True), W.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    ref = O.sparse_conv(xr, Wr, br, nbr.cpu())
    ref.backward(g.double())
    xd, Wd = x.to(DEV).requires_grad_(True), W.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if bias else None
    out = ops.sparse_conv(xd, Wd, bd, nbr, nbr_t)
    out.backward(g.to(DEV))

    def close(a, r, what):
        err = float((a.detach().cpu().double() - r).abs().max())
        scale = float(r.abs().max()) + 1e-12
        assert err <= 2e-5 * scale, f"{what}: {err:.3e} vs scale {scale:.3e}"
    close(out, ref.detach(), "out")
    close(xd.grad, xr.grad, "grad x")
    close(Wd.grad, Wr.grad, "grad W")
    if bias:
        close(bd.grad, br.grad, "grad bias")


@pytest.mark.parametrize("n,cin,cout,k,stride", [(1100, 32, 64, 3, 1), (800, 64, 64, 2, 2), (500, 96, 32, 3, 1)])
def test_sparse_conv_bf16_matches_oracle_on_rounded_operands(n, cin, cout, k, stride):
    from deepviewagg_amd import ops
    torch.manual_seed(n)
    coords = surface_cloud(n, 14, seed=n + 1)
    nbr, nbr_t = _maps(coords, k, stride)
    x = torch.randn(coords.shape[0], cin).bfloat16()
    W = torch.randn(k ** 3, cin, cout) / np.sqrt(cin * k ** 3 / 4)
    g = torch.randn(nbr.shape[1], cout).bfloat16()
    xr = x.double().requires_grad_(True)
    Wr = W.bfloat16().double().requires_grad_(True)         # the kernel rounds the fp32 master weights to bf16
    ref = O.sparse_conv(xr, Wr, None, nbr.cpu())
    ref.backward(g.double())
    xd, Wd = x.to(DEV).requires_grad_(True), W.to(DEV).requires_grad_(True)
    out = ops.sparse_conv(xd, Wd, None, nbr, nbr_t)
    assert out.dtype == torch.bfloat16
    out.backward(g.to(DEV))
    for a, r, tol in [(out, ref.detach(), 2 ** -8), (xd.grad, xr.grad, 2 ** -8), (Wd.grad, Wr.grad, 1e-5)]:
        err = float((a.detach().cpu().double() - r).abs().max())
        assert err <= tol * (float(r.abs().max()) + 1e-12)    # bf16 outputs: half an ulp of the largest value


def test_sparse_conv_rejects_host_tensors_and_bad_shapes():
    from deepviewagg_amd import ops, _lib
    with pytest.raises(_lib.DvaError):
        ops.voxel_kernel_map(torch.zeros(4, 4, dtype=torch.int32), torch.zeros(4, 4, dtype=torch.int32), [[0, 0, 0]])
    coords = surface_cloud(100, 8, seed=0)
    nbr, nbr_t = _maps(coords, 3, 1)
    with pytest.raises(AssertionError):
        ops.sparse_conv(torch.randn(coords.shape[0], 8, device=DEV), torch.randn(27, 16, 16, device=DEV), None,
                        nbr, nbr_t)


def _twin_forward_backward(mods, feats, coords, dev):
    """ResNetDown -> ResNetUp forward + backward; also returns the ReLU activity patterns of the fused BN-ReLU
    layers (to detect pre-activations that straddle zero between two evaluations)."""
    from deepviewagg_amd.modules.SparseConv3d import nn as snn
    x = snn.SparseVoxelTensor(feats.detach().clone().to(dev).requires_grad_(True), coords.to(dev))
    down, up = mods
    active, handles = [], []
    for top in mods:
        for seq in top.modules():
            if isinstance(seq, snn.Seq):
                layers = list(seq)
                for i, m in enumerate(layers[:-1]):
                    if isinstance(m, snn.BatchNorm) and isinstance(layers[i + 1], snn.ReLU):
                        handles.append(m.register_forward_hook(
                            lambda mod, inp, out: active.append((out.F.detach() > 0).cpu())))
    y = down(x)
    z = up(y, x)
    loss = z.F.float().square().mean() + y.F.float().mean()
    loss.backward()
    for hd in handles:
        hd.remove()
    return x, y, z, active


@pytest.mark.parametrize("block", ["ResBlock", "BottleneckBlock"])
def test_resnet_stages_match_cpu_twin(block, monkeypatch):
    """ResNetDown -> ResNetUp (strided conv, residual blocks, transposed conv, skip concatenation, BatchNorm in
    training mode) on the GPU against the same modules evaluated on the CPU with the oracle convolution in fp64.
    A ReLU whose pre-activation is within rounding of zero may switch between the two evaluations (the 3-term
    split leaves ~1e-5 relative error on the features) and changes the gradient of that unit by 100 %: such
    draws are counted (they must stay rare) and the gradient comparison uses a draw without any."""
    from deepviewagg_amd.modules.SparseConv3d import ResNetDown, ResNetUp
    from deepviewagg_amd.modules.SparseConv3d import nn as snn
    coords = surface_cloud(2500, 16, seed=11, batches=2)
    for attempt in range(6):
        torch.manual_seed(3 + attempt)
        feats = torch.randn(coords.shape[0], 16)
        down = ResNetDown(down_conv_nn=[16, 32], N=2, block=block)
        up = ResNetUp(up_conv_nn=[32, 16, 24], N=1, block=block)
        gd, gu = copy.deepcopy(down).to(DEV), copy.deepcopy(up).to(DEV)
        d64, u64 = copy.deepcopy(down).double(), copy.deepcopy(up).double()
        xg, yg, zg, act_g = _twin_forward_backward((gd, gu), feats, coords, DEV)
        with monkeypatch.context() as m:
            m.setattr(snn, "ops", O.OracleOps)
            m.setattr(snn, "batchnorm_act_rows", O.batchnorm_act_rows)
            xc, yc, zc, _ = _twin_forward_backward((down, up), feats, coords, "cpu")             # fp32 on the CPU
            xr, yr, zr, act_r = _twin_forward_backward((d64, u64), feats.double(), coords, "cpu")  # fp64 reference
        assert torch.equal(yg.C.cpu(), yr.C) and torch.equal(zg.C.cpu(), zr.C) and yg.s == 2 and zg.s == 1

        def close(a, c, r, what, tol):
            """GPU error against the fp64 reference: within `tol` of the reference scale, or no worse than 3x
            the error the fp32 CPU evaluation of the same formulas makes."""
            r = r.detach().double()
            err = float((a.detach().cpu().double() - r).abs().max())
            err32 = float((c.detach().double() - r).abs().max())
            assert err <= max(tol * (float(r.abs().max()) + 1e-12), 3.0 * err32), \
                f"{what}: {err:.3e} (cpu fp32 {err32:.3e})"
        close(yg.F, yc.F, yr.F, "encoder features", 1e-4)
        close(zg.F, zc.F, zr.F, "decoder features", 1e-4)
        flips = sum(int((a != b).sum()) for a, b in zip(act_g, act_r))
        assert len(act_g) == len(act_r) > 0 and flips <= 4, f"{flips} ReLU units switched"
        if flips:
            continue
        close(xg.F.grad, xc.F.grad, xr.F.grad, "input gradient", 2e-4)
        names = [n for n, _ in list(gd.named_parameters()) + list(gu.named_parameters())]
        for name, pg, pc, pr in zip(names, list(gd.parameters()) + list(gu.parameters()),
                                    list(down.parameters()) + list(up.parameters()),
                                    list(d64.parameters()) + list(u64.parameters())):
            close(pg.grad, pc.grad, pr.grad, name, 2e-4)
        for (name, bg), bc, br in zip(gd.named_buffers(), down.buffers(), d64.buffers()):
            if name.endswith("num_batches_tracked"):      # the functional oracle does not count; nn.BatchNorm1d does
                assert int(bg) == 1
            else:
                close(bg, bc, br, name, 1e-5)
        return
    pytest.fail("every draw had a ReLU unit within rounding of zero")


def test_sparse_conv_adjoint_property_at_scale():
    """<conv(x), y> == <x, conv^T(y)> on 200k voxels, 64 -> 64 channels (input-gradient kernel against the forward
    kernel) and the weight gradient against the same contraction written with torch index ops on the device."""
    from deepviewagg_amd import ops
    torch.manual_seed(0)
    coords = surface_cloud(400000, 260, seed=5)
    nbr, nbr_t = _maps(coords, 3, 1)
    n = coords.shape[0]
    x = torch.randn(n, 64, device=DEV, requires_grad=True)
    W = (torch.randn(27, 64, 64, device=DEV) / 20).requires_grad_(True)
    y = torch.randn(n, 64, device=DEV)
    out = ops.sparse_conv(x, W, None, nbr, nbr_t)
    out.backward(y)
    lhs = float((out.detach().double() * y.double()).sum())
    rhs = float((x.detach().double() * x.grad.double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * abs(lhs) + 1e-3      # both sides carry the ~2^-16 error of the 3-term split
    k = 5
    dst = torch.nonzero(nbr[k] >= 0).flatten()
    ref = x.detach()[nbr[k][dst].long()].double().t() @ y[dst].double()
    # (3-term split: the dropped lo*lo products leave ~2^-16 relative error per term of the 400k-term sums)
    assert float((W.grad[k].double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert float((W.grad[13].double() - x.detach().double().t() @ y.double()).abs().max()) \
        <= 1e-4 * float((x.detach().double().t() @ y.double()).abs().max())
