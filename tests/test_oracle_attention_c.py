"""The C + OpenMP twin of the view gather + attention tail (oracle/attention_oracle.c, the CPU baseline bench.py
times on the host cores) against the PyTorch restatement, which the reference's golden vectors pin
(tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import attention_oracle as A
from oracle import pooling_oracle as O


@pytest.mark.parametrize("N,C,G,gating,scaling", [(300, 64, 4, True, True), (200, 7, 2, True, False),
                                                  (150, 16, 1, False, True), (100, 32, 32, True, True)])
def test_c_twin_matches_pytorch_oracle(N, C, G, gating, scaling):
    gen = torch.Generator().manual_seed(N + C)
    sizes = torch.randint(0, 9, (N,), generator=gen)
    sizes[:3] = 40
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V, R = int(csr[-1]), 97
    rows = torch.randn(R, C, generator=gen, requires_grad=True)
    row_idx = torch.randint(0, R, (V,), generator=gen)
    compat = torch.randn(V, G, generator=gen, requires_grad=True)
    compat.data[csr[1]:csr[1] + 2] = compat.data[csr[1]]            # a tie: the first maximal view wins
    gate = O.Gating(G) if gating else None
    if gate is not None:
        with torch.no_grad():
            gate.weight.copy_(torch.randn(1, G, generator=gen))
            gate.bias.copy_(torch.randn(1, G, generator=gen))
    w = torch.randn(N, C, generator=gen)
    out, att, gt = O.attention_tail(rows[row_idx], compat, csr, gate, G, C, scaling)
    params = list(gate.parameters()) if gate is not None else []
    grads = torch.autograd.grad((out * w).sum(), [rows, compat] + params)
    gw = gate.weight.detach().view(-1).numpy() if gating else None
    gb = gate.bias.detach().view(-1).numpy() if gating else None
    o2, a2, g2, amax = A.forward(rows.detach().numpy(), row_idx.numpy(), compat.detach().numpy(), csr.numpy(), gw, gb,
                                 scaling)
    np.testing.assert_allclose(o2, out.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(a2, att.detach().numpy(), rtol=1e-4, atol=1e-6)
    if gating:
        np.testing.assert_allclose(g2, gt.detach().view(N, G).numpy(), rtol=1e-4, atol=1e-6)
    gr, gc, gww, gbb = A.backward(w.numpy(), rows.detach().numpy(), row_idx.numpy(), compat.detach().numpy(),
                                  csr.numpy(), a2, g2, amax, gw, gb, scaling)
    np.testing.assert_allclose(gr, grads[0].numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(gc, grads[1].numpy(), rtol=1e-3, atol=1e-5)
    if gating:
        np.testing.assert_allclose(gww, grads[2].view(-1).numpy(), rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(gbb, grads[3].view(-1).numpy(), rtol=1e-3, atol=1e-4)
    assert A.num_threads() >= 1
