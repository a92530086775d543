"""The C + OpenMP twin of DeepSetFeat + E_score (oracle/deepset_oracle.c, part of the CPU baseline bench.py times on the
host cores) against the PyTorch restatement, which the reference's golden vectors pin (tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import deepset_oracle as DS
from oracle import pooling_oracle as O


@pytest.mark.parametrize("N,G,use_num", [(400, 4, True), (250, 2, False), (120, 1, True)])
def test_c_twin_matches_pytorch_oracle(N, G, use_num):
    gen = torch.Generator().manual_seed(7 * N + G)
    sizes = torch.randint(0, 9, (N,), generator=gen)
    sizes[:2] = 45                                       # long points
    sizes[5:8] = 0                                       # points without a view: pooled = 0
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    e_map = O.DeepSetFeat(8, 32, use_num=use_num).train()
    lin = torch.nn.Linear(32, G)
    with torch.no_grad():
        for p in list(e_map.parameters()) + list(lin.parameters()):
            p.copy_(torch.randn(p.shape, generator=gen) * 0.4)
    x_map = torch.rand(V, 8, generator=gen)
    w = torch.randn(V, G, generator=gen)
    s_ref = lin(e_map(x_map, csr))
    g_ref = torch.autograd.grad((s_ref * w).sum(), list(e_map.parameters()) + list(lin.parameters()))
    names = [n for n, _ in e_map.named_parameters()]

    P = DS.params_from_state_dict({k: v.detach().numpy() for k, v in e_map.state_dict().items()},
                                  lin.weight.detach().numpy(), lin.bias.detach().numpy())
    scores, cache = DS.forward(P, x_map.numpy(), csr.numpy(), use_num)
    np.testing.assert_allclose(scores, s_ref.detach().numpy(), rtol=2e-4, atol=2e-4)
    grads = DS.backward(P, cache, w.numpy())
    ref = dict(zip(names, g_ref[:len(names)]))
    for b in DS.BLOCKS:
        dW, dg, db = grads[b]
        for got, key in ((dW, b + ".0.weight"), (dg, b + ".1.batch_norm.weight"), (db, b + ".1.batch_norm.bias")):
            r = ref[key].numpy()
            assert np.abs(got - r).max() <= 2e-3 * np.abs(r).max() + 1e-5, key
    np.testing.assert_allclose(grads["Ws"], g_ref[-2].numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(grads["bs"], g_ref[-1].numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("R,V,K,Co", [(300, 2500, 16, 24), (64, 64, 64, 64)])
def test_c_emod_on_map_rows_matches_per_view_pytorch(R, V, K, Co):
    """E_mod hoisted to the map rows (count-weighted train-mode BatchNorm) + the fusion concat of the C twin against the
    reference's dataflow in PyTorch: gather the rows per view, E_mod on the [V, C] tensor (pooling.py:245,275), autograd."""
    gen = torch.Generator().manual_seed(R + V)
    row_idx = torch.randint(0, R, (V,), generator=gen)
    row_idx[: min(R, V) // 2] = torch.arange(min(R, V) // 2)           # some rows certainly read, some never
    counts = torch.bincount(row_idx, minlength=R).float()
    rows = torch.randn(R, K, generator=gen, requires_grad=True)
    e_mod = O.MLP([K, Co, Co], bias=False).train()
    with torch.no_grad():
        for p in e_mod.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.5)
    w = torch.randn(V, Co, generator=gen)
    out_ref = e_mod(rows[row_idx])                                      # [V, Co]
    g_ref = torch.autograd.grad((out_ref * w).sum(), [rows] + list(e_mod.parameters()))
    # the gradient the hoisted form receives on its output rows: the sum over the views of a row
    d_rows_out = torch.zeros(R, Co).index_add_(0, row_idx, w)

    P = DS.emod_params_from_state_dict({k: v.detach().numpy() for k, v in e_mod.state_dict().items()})
    a, cache = DS.emod_forward(P, rows.detach().numpy(), counts.numpy())
    np.testing.assert_allclose(a[row_idx.numpy()], out_ref.detach().numpy(), rtol=2e-4, atol=2e-4)
    grads, d_rows = DS.emod_backward(P, cache, d_rows_out.numpy())
    names = [n for n, _ in e_mod.named_parameters()]
    ref = dict(zip(names, g_ref[1:]))
    for i in range(2):
        for got, key in zip(grads[i], (f"{i}.0.weight", f"{i}.1.batch_norm.weight", f"{i}.1.batch_norm.bias")):
            r = ref[key].numpy()
            assert np.abs(got - r).max() <= 2e-3 * np.abs(r).max() + 1e-5, key
    r = g_ref[0].numpy()
    assert np.abs(d_rows - r).max() <= 2e-3 * np.abs(r).max() + 1e-5

    x3d, xpool = torch.randn(50, 4, generator=gen), torch.randn(50, Co, generator=gen)
    fused = DS.fusion_concat_forward(x3d.numpy(), xpool.numpy())
    assert np.array_equal(fused, torch.cat([x3d, xpool], 1).numpy())
    d3, dp = DS.fusion_concat_backward(fused, 4)
    assert np.array_equal(d3, x3d.numpy()) and np.array_equal(dp, xpool.numpy())
