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
